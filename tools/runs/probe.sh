cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
(cd tools/exp && timeout -k 10 300 ./mfma_bits 500000 384 5) > gpurun_out/probe/mfma_c2.txt 2>&1 && \
(cd tools/exp && timeout -k 10 120 ./mfma_bits 50000 96 20) > gpurun_out/probe/mfma_c1.txt 2>&1
cat gpurun_out/probe/mfma_c2.txt gpurun_out/probe/mfma_c1.txt
