# How many waves' worth of chains a stage of the latency plan may have before its jobs take more than one term per chain
# (engine.cpp Assembler::msm `room`): run with a build that reads AFX_CHAIN_ROOM there (commit history: the experiment line was
# removed once 4 waves per SIMD = 4096 was chosen).  1 = always four terms per chain, 1000000 = always one.
for room in 1 2048 3072 4096 6144 1000000; do
  echo "== AFX_CHAIN_ROOM=$room"
  for a in "64 16" "32 16" "32 64" "64 256" "8 1024"; do AFX_CHAIN_ROOM=$room python3 tools/mixed_concurrency.py $a | grep -v "kernels of" | cut -c1-110; done
  AFX_CHAIN_ROOM=$room python3 tools/midsize_host_calls.py
done
