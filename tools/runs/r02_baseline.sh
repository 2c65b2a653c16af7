set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
(cd tools/exp && timeout -k 10 300 ./mfma_bits 500000 384 5) > gpurun_out/probe/mfma_c2.txt 2>&1 && \
(cd tools/exp && timeout -k 10 120 ./mfma_bits 50000 96 20) > gpurun_out/probe/mfma_c1.txt 2>&1 && \
BENCH_ARGS="--views 500000 --sensor 128 --headings 32 --agent-steps 0" PROFILE_STEPS=100 timeout -k 10 900 bash tools/profile_bench.sh r02_c2_before > gpurun_out/probe/profile.log 2>&1
echo rc=$?
