#!/bin/bash
# File -> file through the C hosts on tmpfs with the phases on stderr (NAF_GPU_CLI_TIMING), beside the reference:
#   tools/cli_e2e.sh [bytes of FASTA]      (GPU box, repo root; writes gpurun_out/cli_e2e.log)
size=${1:-4e9}
d=/dev/shm/naf_e2e_$$; mkdir -p $d; export TMPDIR=$d
python - "$size" "$d" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from naf_amd import synth
t = synth.fasta_acgt_device(int(float(sys.argv[1])), n_records=24, width=80, seed=5, device="cuda")
t.cpu().numpy().tofile(sys.argv[2] + "/a.fa")
PY
ls -l $d/a.fa
TIMEFORMAT="   wall %R s user %U sys %S"
for i in 1 2; do
  echo "ennaf file -> file"; time NAF_GPU_CLI_TIMING=1 naf_amd/bin/ennaf $d/a.fa -o $d/a.naf
  echo "unnaf file -> file"; time NAF_GPU_CLI_TIMING=1 naf_amd/bin/unnaf $d/a.naf -o $d/a.out
done
cmp $d/a.fa $d/a.out && echo roundtrip ok
echo "unnaf -c > /dev/null"; time naf_amd/bin/unnaf -c $d/a.naf > /dev/null
echo "cat | ennaf -c > file"; time sh -c "cat $d/a.fa | naf_amd/bin/ennaf -c > $d/b.naf"
cmp $d/a.naf $d/b.naf && echo pipe archive identical
if [ -x oracle/_ref/unnaf ]; then
  echo "reference unnaf"; time oracle/_ref/unnaf $d/a.naf -o $d/r.out
  echo "reference ennaf"; time oracle/_ref/ennaf $d/a.fa -o $d/r.naf
fi
rm -rf $d
