"""GPU tests of the drop-in boundary: the C hosts ennaf/unnaf (naf_amd/bin) run the reference's own
test matrix (tests/golden/ref_tests, ref_cases.json) and interoperate with the real reference binaries."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, golden_bytes, naf_cases, ref_cases

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "naf_amd", "bin")


def pipe(text, eargs, uargs):
    e = subprocess.run([os.path.join(BIN, "ennaf"), *eargs, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    u = subprocess.run([os.path.join(BIN, "unnaf"), *uargs, "-c"], input=e.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return e, u


@pytest.mark.parametrize("case", ref_cases(), ids=lambda c: c["set"] + "/" + c["name"])
def test_reference_test_matrix_through_the_clis(case):
    text = golden_bytes("ref_tests", case["set"], case["input"])
    eargs = [a for a in case["ennaf_args"] if a not in ("-22",)]
    if "--long" in eargs:
        i = eargs.index("--long"); del eargs[i:i + 2]
    e, u = pipe(text, eargs, case["unnaf_args"])
    pre = os.path.join(GOLDEN, "ref_tests", case["set"], case["name"])
    assert e.returncode == 0 and u.returncode == 0, (e.stderr, u.stderr)
    assert u.stdout == open(pre + ".out-ref", "rb").read()
    assert e.stderr == open(pre + ".e.err-ref", "rb").read()
    assert u.stderr == open(pre + ".u.err-ref", "rb").read()


def test_interop_with_the_real_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        # reference-made archive -> our unnaf CLI == reference unnaf
        mine = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert mine.returncode == 0 and mine.stdout == oracle.ref_unnaf(naf), case["name"]
        # our ennaf CLI -> reference unnaf == reference ennaf -> reference unnaf
        fq = bool(h.flags & 1)
        text = oracle.ref_unnaf(naf, ("--fastq",) if fq else ("--fasta",))
        args = [a for a in case["ennaf_args"] if not a.startswith("-1") and a not in ("-19", "-3")]
        if "--long" in args:
            i = args.index("--long"); del args[i:i + 2]
        e = subprocess.run([os.path.join(BIN, "ennaf"), *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert e.returncode == 0, e.stderr
        assert oracle.ref_unnaf(e.stdout, ("--fastq",) if fq else ("--fasta",)) == text, case["name"]
        if fq:
            continue
        for m in ("--ids", "--names", "--lengths", "--mask", "--total-length", "--number"):
            if m == "--mask" and "--no-mask" in args:
                continue
            a = subprocess.run([os.path.join(BIN, "unnaf"), m, "-c"], input=naf, stdout=subprocess.PIPE, timeout=120).stdout
            assert a == oracle.ref_unnaf(naf, (m,)), (case["name"], m)
