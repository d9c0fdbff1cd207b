# the C driver on the 0.79 GB headline text (BASELINE configs[2] stand-in) in /dev/shm: -s and -S, outputs through the files'
# mapped pages (default) against the pinned-buffer + pwrite path (PFP_MAP_OUTPUT=0).  GPU box: bash tools/microbench/cli_c3.sh
python - <<'P'
import sys, importlib
sys.path.insert(0, ".")
import __graft_entry__ as e
e.load_package()
import torch
synth = importlib.import_module("bigbwt_amd.synth")
t = synth.workload_text_torch(torch.device("cuda", 0), "c3")
t.cpu().numpy().tofile("/dev/shm/c3.fa")
P
sleep 5
for flag in ${CLI_C3_FLAGS:--s -S}; do
for envs in ${CLI_PROBE_ENVS:-X=1 PFP_MAP_OUTPUT=0 X=2 PFP_MAP_OUTPUT=0}; do
  echo "=== $flag $envs"
  s=$(date +%s%N)
  env $envs PFP_TRACE_HOST=1 big-bwt_amd/bigbwt $flag /dev/shm/c3.fa 2>&1 | grep "file to files\|Total construction\|ctx_destroy\|through its mapping"
  e=$(date +%s%N)
  echo "process $(( (e - s) / 1000000 )) ms"
  cat /dev/shm/c3.fa.bwt /dev/shm/c3.fa.s* | sha256sum | cut -c1-16
  rm -f /dev/shm/c3.fa.*
  sleep 2
done
done
rm -f /dev/shm/c3.fa*
