"""Deterministic synthetic FASTA/FASTQ generators (SURVEY.md section 8(d) workloads).

CPU generators use numpy (tests, fixtures); `fasta_acgt_device` builds the multi-GB bench inputs
directly in HBM with torch so bench.py never round-trips 10 GB through the host.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def wrap_lines(seq: np.ndarray, width: int) -> bytes:
    """seq bytes -> text with '\n' after every `width` bases and at the end (none if empty)."""
    n = len(seq)
    if n == 0:
        return b""
    if width <= 0:
        return seq.tobytes() + b"\n"
    full, rem = divmod(n, width)
    out = np.empty(n + full + (1 if rem else 0), dtype=np.uint8)
    if full:
        body = out[: full * (width + 1)].reshape(full, width + 1)
        body[:, :width] = seq[: full * width].reshape(full, width)
        body[:, width] = 10
    if rem:
        out[full * (width + 1): full * (width + 1) + rem] = seq[full * width:]
        out[-1] = 10
    return out.tobytes()


def fasta_acgt(n_bases: int, n_records: int = 1, width: int = 80, seed: int = 12345,
               name=lambda k: b">seq%d synthetic random ACGT" % (k + 1)) -> bytes:
    """cfg1-style: uniform ACGT, `n_records` records of (almost) equal length, `width`-column lines."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    per = n_bases // n_records
    for k in range(n_records):
        n = per if k + 1 < n_records else n_bases - per * (n_records - 1)
        seq = _ACGT[rng.integers(0, 4, n, dtype=np.uint8)]
        parts.append(name(k) + b"\n" + wrap_lines(seq, width))
    return b"".join(parts)


def fasta_mixed(n_records: int = 20, mean_len: int = 3000, width: int = 60, seed: int = 1,
                p_lower: float = 0.3, iupac: bool = True, empty_every: int = 7) -> bytes:
    """Soft-masked runs, N runs, IUPAC codes, empty records -- exercises mask RLE and emit edge cases."""
    rng = np.random.Generator(np.random.PCG64(seed))
    alpha = np.frombuffer(b"ACGT" * 8 + (b"RYSWKMBDHVN-" if iupac else b""), dtype=np.uint8)
    parts = []
    for k in range(n_records):
        if empty_every and k % empty_every == empty_every - 1:
            n = 0
        else:
            n = int(rng.integers(1, 2 * mean_len))
        seq = alpha[rng.integers(0, len(alpha), n)]
        pos = 0
        while pos < n:                                   # alternate upper / lower runs
            run = int(rng.geometric(1.0 / 300))
            if rng.random() < p_lower:
                seg = seq[pos:pos + run]
                letters = seg != ord("-")
                seg[letters] |= 0x20
            pos += run
        if n > 600 and rng.random() < 0.5:
            a = int(rng.integers(0, n - 520))
            seq[a:a + 510] = ord("N")
        hdr = b">rec%d" % k + (b" some comment %d" % (k * 7) if k % 3 else b"")
        parts.append(hdr + b"\n" + wrap_lines(seq, width))
    return b"".join(parts)


def fastq_reads(n_reads: int, read_len: int = 150, seed: int = 7, var_len: bool = False) -> bytes:
    """cfg5-style: ACGT 0.22 each, acgt 0.025 each, N 0.02; quality uniform Phred 0-40."""
    rng = np.random.Generator(np.random.PCG64(seed))
    alpha = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    p = np.array([0.22] * 4 + [0.025] * 4 + [0.02])
    parts = []
    for k in range(n_reads):
        n = int(rng.integers(1, read_len + 1)) if var_len else read_len
        seq = alpha[rng.choice(9, n, p=p)]
        qual = rng.integers(33, 74, n, dtype=np.uint8)
        parts.append(b"@read%d len=%d\n" % (k + 1, n) + seq.tobytes() + b"\n+\n" + qual.tobytes() + b"\n")
    return b"".join(parts)


def fasta_acgt_device(total_bytes: int, n_records: int = 100, width: int = 80, seed: int = 2024, device="cuda"):
    """cfg2/cfg3-style FASTA of ~total_bytes built in device memory; returns a uint8 torch tensor.

    Every record is `>chrK synthetic random ACGT record K\n` + bases wrapped at `width`.  Record
    lengths are multiples of `width` (so the text is exactly header + rows of width+1 bytes).
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hdrs = [b">chr%d synthetic random ACGT record %d\n" % (k + 1, k + 1) for k in range(n_records)]
    hdr_total = sum(len(h) for h in hdrs)
    rows_total = max(n_records, (total_bytes - hdr_total) // (width + 1))
    rows_per = rows_total // n_records
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    total = hdr_total + rows_per * n_records * (width + 1)
    out = torch.empty(total, dtype=torch.uint8, device=device)
    pos = 0
    for k in range(n_records):
        h = torch.tensor(list(hdrs[k]), dtype=torch.uint8, device=device)
        out[pos:pos + len(h)] = h
        pos += len(h)
        body = out[pos:pos + rows_per * (width + 1)].view(rows_per, width + 1)
        idx = torch.randint(0, 4, (rows_per, width), dtype=torch.int64, device=device, generator=g)
        body[:, :width] = lut[idx]
        body[:, width] = 10
        pos += rows_per * (width + 1)
    return out


def fastq_reads_device(total_bytes: int, read_len: int = 150, seed: int = 7, device="cuda", chunk_reads: int = 1 << 22):
    """cfg5-style FASTQ of ~total_bytes built in device memory (uint8 torch tensor).

    Reads `@readN len=L`, bases ACGT 0.22 each / acgt 0.025 each / N 0.02, quality uniform Phred 0-40.
    Reads are generated in groups that share the number of digits of N, so that every group is a
    fixed-width 2-D byte matrix.
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base_lut = torch.tensor(list(b"A" * 220 + b"C" * 220 + b"G" * 220 + b"T" * 220 + b"a" * 25 + b"c" * 25 + b"g" * 25 + b"t" * 25 + b"N" * 20),
                            dtype=torch.uint8, device=device)
    tail = b" len=%d\n" % read_len
    parts, made, n = [], 0, 1
    while made < total_bytes:
        digits = len(str(n))
        width = 5 + digits + len(tail) + read_len + 3 + read_len + 1
        cnt = min(10 ** digits - n, chunk_reads, max(1, (total_bytes - made + width - 1) // width))
        m = torch.empty((cnt, width), dtype=torch.uint8, device=device)
        m[:, 0:5] = torch.tensor(list(b"@read"), dtype=torch.uint8, device=device)
        ids = torch.arange(n, n + cnt, dtype=torch.int64, device=device)
        for d in range(digits):
            m[:, 5 + d] = ((ids // (10 ** (digits - 1 - d))) % 10 + 48).to(torch.uint8)
        p = 5 + digits
        m[:, p:p + len(tail)] = torch.tensor(list(tail), dtype=torch.uint8, device=device)
        p += len(tail)
        m[:, p:p + read_len] = base_lut[torch.randint(0, 1000, (cnt, read_len), dtype=torch.int64, device=device, generator=g)]
        p += read_len
        m[:, p:p + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=device)
        p += 3
        m[:, p:p + read_len] = torch.randint(33, 74, (cnt, read_len), dtype=torch.int64, device=device, generator=g).to(torch.uint8)
        m[:, p + read_len] = 10
        parts.append(m.reshape(-1))
        made += cnt * width
        n += cnt
    return torch.cat(parts)


def repeat_genome(seed=3, unit=40000, copies=50):
    """Repeat-rich FASTA: `copies` mutated copies of one random unit (150 substitutions each, every third with a soft-masked
    stretch) separated by random filler of 10..4000 bases -- what the match finders of the higher levels and --long are for."""
    rng = np.random.Generator(np.random.PCG64(seed))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = acgt[rng.integers(0, 4, unit)]
    parts = []
    for i in range(copies):
        s = base.copy()
        idx = rng.integers(0, unit, 150)
        s[idx] = acgt[rng.integers(0, 4, 150)]
        if i % 3 == 0:
            s[1000:3000] |= 0x20
        parts.append(s)
        parts.append(acgt[rng.integers(0, 4, int(rng.integers(10, 4000)))])
    seq = np.concatenate(parts)
    third = len(seq) // 3
    out = b""
    for k in range(3):
        out += b">rep%d repeat-rich synthetic\n" % k + wrap_lines(seq[k * third:(k + 1) * third], 60)
    return out


def softmask_device(text, seed=3, lo=20, hi=600):
    """Soft-mask a device FASTA text in place: alternating upper / lower-case runs of lo..hi-1 bases (uniform), the way repeat-masked
    genome assemblies look (about half of the bases in lower case).  Letters of header lines change case too: harmless for the codec."""
    import torch
    n = text.numel()
    gen = torch.Generator(device=text.device); gen.manual_seed(seed)
    nruns = n // ((lo + hi) // 2 - 60) + 16
    bounds = torch.cumsum(torch.randint(lo, hi, (nruns,), device=text.device, generator=gen), 0)
    CH = 1 << 28
    for s in range(0, n, CH):
        e = min(n, s + CH)
        pos = torch.arange(s, e, device=text.device)
        par = torch.searchsorted(bounds, pos, right=True) & 1
        seg = text[s:e]
        seg[((seg >= 65) & (seg <= 90)) & (par == 1)] += 32
    return text


def realistic_genome_device(total_bytes: int, n_records: int = 24, width: int = 60, seed: int = 11, device="cuda",
                            gc: float = 0.41, cpg_keep: float = 0.22, n_run_every: int = 40_000_000, iupac_every: int = 1_000_000,
                            softmask: bool = True):
    """What an assembled, repeat-masked vertebrate genome looks like to the codec, built in device memory (uint8 torch tensor):
    GC content `gc`, CpG depleted to `cpg_keep` of its expected frequency (a C followed by G is mostly rewritten to CA -- the pair
    histogram of the packed stream is visibly skewed), records of unequal length that start and end in a run of N (telomeres) and
    hold a run of 5 k .. 100 k N every `n_run_every` bases on average (gaps), an IUPAC code every `iupac_every` bases, `width`-column
    lines, and -- softmask=True -- alternating upper / lower-case runs of 20..600 bases.  Header lines avoid the letters A C G T."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # record lengths: a geometric-ish spread like chromosomes (largest about five times the smallest)
    w = torch.linspace(5.0, 1.0, n_records, dtype=torch.float64)
    hdrs = [b">REF%d L=%d SYN\n" % (k + 1, k) for k in range(n_records)]
    hdr_total = sum(len(h) for h in hdrs)
    rows_total = max(n_records, (total_bytes - hdr_total) // (width + 1))
    rows = [max(1, int(rows_total * float(x) / float(w.sum()))) for x in w]
    total = hdr_total + sum(rows) * (width + 1)
    out = torch.empty(total, dtype=torch.uint8, device=device)
    at, cg = (1.0 - gc) / 2, gc / 2
    cum = torch.tensor([at, at + cg, at + 2 * cg], dtype=torch.float32, device=device)       # A C G T
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    iupac = torch.tensor(list(b"RYKMSWBDHV"), dtype=torch.uint8, device=device)
    pos = 0
    CH = 1 << 22                                                           # rows per step
    for k in range(n_records):
        h = torch.tensor(list(hdrs[k]), dtype=torch.uint8, device=device)
        out[pos:pos + len(h)] = h
        pos += len(h)
        nb = rows[k] * (width + 1)
        body = out[pos:pos + nb].view(rows[k], width + 1)
        for r0 in range(0, rows[k], CH):
            r1 = min(rows[k], r0 + CH)
            u = torch.rand((r1 - r0) * width, dtype=torch.float32, device=device, generator=g)
            s = lut[torch.bucketize(u, cum)]
            # CpG depletion: most G that follow a C become A
            cpg = (s[:-1] == 67) & (s[1:] == 71) & (torch.rand(s.numel() - 1, dtype=torch.float32, device=device, generator=g) >= cpg_keep)
            s[1:][cpg] = 65
            body[r0:r1, :width] = s.view(r1 - r0, width)
            del u, s, cpg
        body[:, width] = 10
        seg = out[pos:pos + nb]
        # telomeres and gaps: runs of N in text coordinates (line ends stay)
        nbases = rows[k] * width
        runs = [(0, min(nb, 10_000 + 167 * k)), (max(0, nb - 10_000 - 61 * k), nb)]
        n_gaps = int(nbases // n_run_every)
        if n_gaps:
            starts = torch.randint(0, max(1, nb - 200_000), (n_gaps,), device="cpu", generator=torch.Generator().manual_seed(seed * 1000 + k))
            lens = torch.randint(5_000, 100_000, (n_gaps,), device="cpu", generator=torch.Generator().manual_seed(seed * 1000 + k + 500))
            runs += [(int(a), int(a) + int(l)) for a, l in zip(starts.tolist(), lens.tolist())]
        for a, b in runs:
            piece = seg[a:b]
            piece[piece != 10] = 78
        n_iu = int(nbases // iupac_every)
        if n_iu:
            p = torch.randint(0, nb, (n_iu,), device=device, generator=g)
            cur = seg[p]
            ok = (cur == 65) | (cur == 67) | (cur == 71) | (cur == 84)
            seg[p[ok]] = iupac[torch.randint(0, iupac.numel(), (int(ok.sum().item()),), device=device, generator=g)]
        pos += nb
    if softmask:
        softmask_device(out, seed=seed + 1)
    return out


def repeat_genome_device(total_bytes: int, n_records: int = 8, width: int = 60, seed: int = 5, device="cuda",
                         families: int = 96, coverage: float = 0.45, div_lo: float = 0.003, div_hi: float = 0.06):
    """Repeat-rich genome in device memory (uint8 torch tensor): random ACGT of which `coverage` is overwritten by copies of `families`
    repeat units (300 .. 6000 bases, the spread of SINE / LINE fragments), every copy with its own substitution rate between `div_lo` and
    `div_hi` -- what the match finders of levels >= 2 and `--long` are for (ennaf/src/compressor.c:7-21, ennaf.c:247-273).  Copies land at
    any base offset, so half of them are a nibble off their unit in the packed stream: a byte-wise match finder sees the other half."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    hdrs = [b">scaffold%d repeat-rich synthetic\n" % (k + 1) for k in range(n_records)]
    hdr_total = sum(len(h) for h in hdrs)
    rows_per = max(1, (total_bytes - hdr_total) // (width + 1) // n_records)
    nb = rows_per * n_records * width
    seq = torch.randint(0, 4, (nb,), dtype=torch.uint8, device=device, generator=g)
    cpu = torch.Generator().manual_seed(seed * 7 + 1)
    lens = torch.randint(300, 6000, (families,), generator=cpu).tolist()
    per_family = int(nb * coverage / families)
    for f, L in enumerate(lens):
        unit = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=g)
        copies = max(1, per_family // L)
        at = torch.randint(0, nb - L, (copies,), dtype=torch.int64, device=device, generator=g)
        rate = div_lo + (div_hi - div_lo) * torch.rand((copies, 1), device=device, generator=g)
        body = unit.repeat(copies, 1)
        hit = torch.rand((copies, L), device=device, generator=g) < rate
        body[hit] = torch.randint(0, 4, (int(hit.sum().item()),), dtype=torch.uint8, device=device, generator=g)
        idx = at[:, None] + torch.arange(L, device=device)[None, :]
        seq[idx.reshape(-1)] = body.reshape(-1)
        del body, hit, idx
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty(hdr_total + rows_per * n_records * (width + 1), dtype=torch.uint8, device=device)
    pos = 0
    for k in range(n_records):
        h = torch.tensor(list(hdrs[k]), dtype=torch.uint8, device=device)
        out[pos:pos + len(h)] = h; pos += len(h)
        body = out[pos:pos + rows_per * (width + 1)].view(rows_per, width + 1)
        body[:, :width] = lut[seq[k * rows_per * width:(k + 1) * rows_per * width].to(torch.int64)].view(rows_per, width)
        body[:, width] = 10
        pos += rows_per * (width + 1)
    return out
