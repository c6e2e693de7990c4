d=/dev/shm/naf_ab_$$; mkdir -p $d; export TMPDIR=$d
python - "$d" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from naf_amd import synth
t = synth.fasta_acgt_device(int(4e9), n_records=24, width=80, seed=5, device="cuda")
t.cpu().numpy().tofile(sys.argv[1] + "/a.fa")
PY
naf_amd/bin/ennaf $d/a.fa -o $d/a.naf
for i in 1 2 3; do
  echo "--- IO_SMALL=1 (lanes by 64 MiB, 4 MiB chunks)"; NAF_GPU_CLI_TIMING=1 naf_amd/bin/unnaf -c $d/a.naf 2>&1 > /dev/null | grep -E "upload|download|init"
  echo "--- IO_SMALL=0 (lanes by 256 MiB, 16 MiB chunks)"; NAF_GPU_IO_SMALL=0 NAF_GPU_CLI_TIMING=1 naf_amd/bin/unnaf -c $d/a.naf 2>&1 > /dev/null | grep -E "upload|download|init"
done
rm -rf $d
