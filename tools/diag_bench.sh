# builds and runs tools/diag_bench.hip on the GPU box (the 32 x 32 diagonal block routines: cycles and error against a host Cholesky) -> gpurun_out/r05_diag_bench.txt
export TMPDIR=/tmp; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepcgp_amd/csrc -I tools tools/diag_bench.hip -o /tmp/diag_bench 2>/dev/null && /tmp/diag_bench > gpurun_out/r05_diag_bench.txt 2>&1
cat gpurun_out/r05_diag_bench.txt
