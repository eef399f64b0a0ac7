#!/bin/sh
# Round 6 GPU sessions: tools/gpu_r06.sh <stage>   (run through gpurun from the repository root; output under gpurun_out/)
set -x
mkdir -p gpurun_out
case "$1" in
trees)
	python -m pytest tests/test_gpu_trees.py -x -q 2>&1 | tail -25 > gpurun_out/r06_trees.txt
	python -m pytest tests/test_gpu_dropin.py -x -q -k "device_pairs" 2>&1 | tail -8 >> gpurun_out/r06_trees.txt
	python -m pytest tests/test_gpu_world.py tests/test_gpu_broadphase.py -x -q 2>&1 | tail -8 >> gpurun_out/r06_trees.txt
	(tools/dropin_product_demo.sh 200 40 pyramid 3 4 2 45; S2AMD_DEVICE_TREES=0 S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY="$PWD/solver2d_amd/libs2amd.so" tools/dropin_product_demo.bin 200 40 pyramid 3 4 2 45) > gpurun_out/r06_dropin_default_solver.txt 2>&1
	tools/dropin_product_demo.sh 200 40 pyramid 7 8 4 45 > gpurun_out/r06_dropin_demo.txt 2>&1
	cat gpurun_out/r06_trees.txt; tail -30 gpurun_out/r06_dropin_default_solver.txt
	;;
trees2)
	python -m pytest tests/test_gpu_trees.py -x -q 2>&1 | tail -25 > gpurun_out/r06_trees.txt
	python -m pytest tests/test_gpu_dropin.py -x -q -k "device_pairs" 2>&1 | tail -8 >> gpurun_out/r06_trees.txt
	(tools/dropin_product_demo.sh 200 40 pyramid 3 4 2 45; S2AMD_DEVICE_TREES=0 S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY="$PWD/solver2d_amd/libs2amd.so" tools/dropin_product_demo.bin 200 40 pyramid 3 4 2 45) > gpurun_out/r06_dropin_default_solver.txt 2>&1
	cd /tmp && export TMPDIR=/tmp && S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY="$GRAFT_REPO_ROOT/solver2d_amd/libs2amd.so" rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_trees -o trace -- $GRAFT_REPO_ROOT/tools/dropin_product_demo.bin 200 40 pyramid 3 4 2 45 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
	python tools/rocpd_summary.py $(find gpurun_out/prof_trees -name '*_results.db' | head -1) > gpurun_out/r06_trees_kernel_trace.txt 2>&1; rm -rf gpurun_out/prof_trees
	cat gpurun_out/r06_trees.txt; tail -12 gpurun_out/r06_dropin_default_solver.txt; head -40 gpurun_out/r06_trees_kernel_trace.txt
	;;
full)
	python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gpu_suite.txt
	tools/dropin_product_demo.sh 200 40 pyramid 7 8 4 45 > gpurun_out/r06_dropin_demo.txt 2>&1
	S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY="$PWD/solver2d_amd/libs2amd.so" tools/dropin_product_demo.bin 200 40 pyramid 3 4 2 45 >> gpurun_out/r06_dropin_demo.txt 2>&1
	python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
	cat gpurun_out/r06_gpu_suite.txt; tail -4 gpurun_out/r06_dropin_demo.txt; cat gpurun_out/r06_bench.json | cut -c1-1500
	;;
sharded)
	python -m pytest tests/test_gpu_sharded.py tests/test_gpu_reshard.py -x -q 2>&1 | tail -15 > gpurun_out/r06_sharded.txt
	python bench.py --steps 100 --warmup 20 --no-cpu > gpurun_out/r06_bench_sharded.json 2>gpurun_out/r06_bench_sharded.err
	cat gpurun_out/r06_sharded.txt; python -c "import json;d=json.load(open('gpurun_out/r06_bench_sharded.json'));print(json.dumps(d['sharded_abi'],indent=1)[:3000]);print(d['island_sharded']['ms_per_step'])"
	;;
*)
	echo "unknown stage $1"; exit 2;;
esac
