cd $GRAFT_REPO_ROOT
for n in BASE SWZ BASE SWZ; do
  if [ $n = BASE ]; then L=spice_amd/csrc/libspicegpu.so; else L=tools/diag/lib_$n.so; fi
  echo "== $n bench" >> gpurun_out/s4_ab10.log
  timeout -k 10 200 python -u tools/bench_with.py $L --also 0 --no-cpu-baseline --no-end-to-end --verify 2 --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms'], d.get('verified_sessions'), d.get('lone_session_frame_ms'))" >> gpurun_out/s4_ab10.log
done
