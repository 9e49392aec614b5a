mkdir -p gpurun_out/r4b
timeout 300 python -m pytest tests/test_gpu_geometry.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4b/geom.txt; cat gpurun_out/r4b/geom.txt
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4b/gputests.txt; cat gpurun_out/r4b/gputests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4b/bench_default.json 2> gpurun_out/r4b/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4b/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "parity", d["parity"]["ok"], "busy", d["host"]["cpu_cores_busy"])
r=d["roofline"]; print({k:r.get(k) for k in ("exclusive_us_per_frame_all_kernels","ceiling_frames_per_s","value_over_ceiling","exclusive_us")})
PY
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
timeout 120 python bench.py $Q > gpurun_out/r4b/b60.json 2> gpurun_out/r4b/b60.err
ICG_HOST_PROF=cpu ICG_BENCH_TIMED_CPUS=2 timeout 120 python bench.py $Q --groups 8 --streams 768 > gpurun_out/r4b/q2_prof.json 2> gpurun_out/r4b/q2_prof.err
python - <<'PY'
import json
for n in ("b60","q2_prof"):
    try:
        d=json.loads(open("gpurun_out/r4b/%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["host"])
    except Exception as e: print(n,"failed",e)
PY
tail -40 gpurun_out/r4b/q2_prof.err
