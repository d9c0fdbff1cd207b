#!/bin/bash
mkdir -p gpurun_out
python3 - <<'PY'
import importlib, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
entry.load_package()
synth = importlib.import_module("bigbwt_amd.synth")
synth.workload_text_np("c3").tofile("/dev/shm/pfp_t.fa")
PY
for T in 1 8 4 2 8 1; do
  rm -f /dev/shm/pfp_t.fa.*
  PFP_PWRITE_THREADS=$T PFP_TRACE_HOST=1 timeout -k 10 120 big-bwt_amd/bigbwt -w 10 -p 100 -s -e /dev/shm/pfp_t.fa 2>&1 | grep -E "file to files|Total construction" | tr '\n' ' '; echo " threads=$T"
done
sha256sum /dev/shm/pfp_t.fa.bwt | cut -c1-16
rm -f /dev/shm/pfp_t.fa*
