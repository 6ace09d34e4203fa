cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out/r05f
( timeout 600 python -m pytest "tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_headline_batch" -x -q 2>&1 | grep -v amdgpu.ids | grep -v "^$" | head -80 ) > gpurun_out/r05f/pytest.log
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for i in 1 2 3; do CHAM_PIPE_HALVES=0 timeout 300 python bench.py $BARGS > gpurun_out/r05f/b0_$i.out 2> gpurun_out/r05f/b0_$i.err; echo "rc $?" >> gpurun_out/r05f/b0_$i.err; done
head -70 gpurun_out/r05f/pytest.log; for i in 1 2 3; do tail -5 gpurun_out/r05f/b0_$i.err; wc -c gpurun_out/r05f/b0_$i.out; done
