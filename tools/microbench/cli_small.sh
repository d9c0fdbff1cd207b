# the C driver on a 12 MB input (BASELINE configs[0] size): where a cold process spends its time (GPU box)
python - <<'P'
import sys, importlib
sys.path.insert(0, ".")
import __graft_entry__ as e
e.load_package()
synth = importlib.import_module("bigbwt_amd.synth")
synth.workload_text_np("c1").tofile("/dev/shm/c1.fa")
P
for k in 1 2 3; do
  s=$(date +%s%N)
  PFP_TRACE_HOST=1 big-bwt_amd/bigbwt /dev/shm/c1.fa 2>&1 | grep "pfp\]\|Total construction"
  e=$(date +%s%N)
  echo "process $(( (e - s) / 1000000 )) ms"
done
s=$(date +%s%N); big-bwt_amd/bigbwt -h > /dev/null; e=$(date +%s%N); echo "bigbwt -h (process start + library load, no GPU work): $(( (e - s) / 1000000 )) ms"
rm -f /dev/shm/c1.fa*
