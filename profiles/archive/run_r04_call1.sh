bash profiles/run_r04_opencv_probe.sh
mkdir -p gpurun_out/r4a
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4a/gputests.txt; cat gpurun_out/r4a/gputests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err; tail -c 1500 gpurun_out/r4a/bench_default.json
bash profiles/run_r04_first_call.sh
