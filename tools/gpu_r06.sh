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
numbers)
	O=gpurun_out/r6; mkdir -p $O
	timeout 900 python bench.py > $O/r06_bench_final.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $O/r06_bench_final.json
	timeout 600 python tools/solver_table.py --steps 100 > $O/r06_solver_table.jsonl 2> $O/solver_table.err
	timeout 300 python tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS,TGS_Soft,PGS_NGS_Block --steps 100 >> $O/r06_solver_table.jsonl 2>> $O/solver_table.err
	timeout 300 python tools/config3_bench.py Jacobi 2> $O/config3.err | tail -1 > $O/r06_config3_tumbler_jacobi.json
	timeout 300 python tools/config3_bench.py TGS_Soft 2>> $O/config3.err | tail -1 > $O/r06_config3b_tumbler_tgs_soft.json
	timeout 300 python tools/churn_bench.py --world tumbler --steps 200 > $O/r06_churn_tumbler.json 2> $O/churn_tumbler.err
	(tools/dropin_product_demo.sh 200 40 pyramid 7 8 4 45; tools/dropin_product_demo.sh 200 40 pyramid 3 4 2 45; S2AMD_DEVICE_TREES=0 S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY=$PWD/solver2d_amd/libs2amd.so tools/dropin_product_demo.bin 200 40 pyramid 3 4 2 45; tools/dropin_product_demo.sh 10000 40 tumbler 0 4 2 120) > $O/r06_dropin_demo.txt 2>&1
	python bench.py --sharded-abi-devices 0 > $O/r06_sharded_abi_one_device.json 2>/dev/null
	cut -c1-220 $O/r06_solver_table.jsonl; cut -c1-300 $O/r06_config3_tumbler_jacobi.json $O/r06_config3b_tumbler_tgs_soft.json; cat $O/r06_dropin_demo.txt; cut -c1-400 $O/r06_sharded_abi_one_device.json
	;;
*)
	echo "unknown stage $1"; exit 2;;
esac
