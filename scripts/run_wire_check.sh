set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wire or key_table" > gpurun_out/wire_keys_test.log 2>&1 || { tail -30 gpurun_out/wire_keys_test.log; exit 1; }
tail -3 gpurun_out/wire_keys_test.log
timeout -k 10 300 python bench.py --wire --steps 10 --warmup 3 > gpurun_out/bench_wire_new.json 2> gpurun_out/bench_wire_new.err
cat gpurun_out/bench_wire_new.json
