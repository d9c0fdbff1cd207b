# the C driver alone on the 12.6 GB north-star text in /dev/shm (GPU box): where a cold process spends its time, and what the
# host-side knobs change.  usage: bash tools/microbench/cli_probe.sh
python - <<'P'
import sys, importlib
sys.path.insert(0, ".")
import __graft_entry__ as e
e.load_package()
import torch
synth = importlib.import_module("bigbwt_amd.synth")
t = synth.workload_text_torch(torch.device("cuda", 0), "huge_s")
with open("/dev/shm/probe.fa", "wb") as fh:
    for s in range(0, t.numel(), 1 << 28):
        fh.write(t[s:s + (1 << 28)].cpu().numpy().tobytes())
P
ls -la /dev/shm/probe.fa
sleep 20
for envs in ${CLI_PROBE_ENVS:-X=1 PFP_MAP_OUTPUT=0 X=2}; do
  echo "=== $envs"
  s=$(date +%s%N)
  env $envs PFP_TRACE_HOST=1 big-bwt_amd/bigbwt -s /dev/shm/probe.fa 2>&1 | grep "pfp\]\|Total construction"
  e=$(date +%s%N)
  echo "process $(( (e - s) / 1000000 )) ms"
  sha256sum /dev/shm/probe.fa.bwt /dev/shm/probe.fa.ssa | cut -c1-16
  rm -f /dev/shm/probe.fa.bwt /dev/shm/probe.fa.ssa /dev/shm/probe.fa.log
  sleep 5
done
rm -f /dev/shm/probe.fa*
