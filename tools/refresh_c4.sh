# C4 evidence for profiles/: per-kernel rocprofv3 stats at 8 and 16 heads and the compiled-out-parts timing harnesses.
# Run on the GPU box:  gpurun -- 'bash tools/refresh_c4.sh'   (the harness binaries are built beforehand, BUILD_ONLY=1)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4ref
VARIANTS="0 1 2 4 6 8 14 16 32" timeout -k 5 300 tools/micro/enc_bench.sh > gpurun_out/c4ref/harness.txt
VARIANTS="0:0 1:1 2:2 8:8 16:0 0:3" timeout -k 5 400 tools/micro/attn_bench.sh >> gpurun_out/c4ref/harness.txt
echo harness done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4ref/h8 -o c4 -- python3 bench.py --workload c4 --steps 50 --cpu-seconds 0 > gpurun_out/c4ref/c4_h8_prof.json 2> gpurun_out/c4ref/h8.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4ref/h16 -o c4 -- python3 bench.py --workload c4 --heads 16 --steps 50 --cpu-seconds 0 > gpurun_out/c4ref/c4_h16_prof.json 2> gpurun_out/c4ref/h16.err
echo stats done
timeout -k 5 200 python bench.py --workload c4 --steps 100 --cpu-seconds 0 > gpurun_out/c4ref/c4_h8.json
timeout -k 5 200 python bench.py --workload c4 --heads 16 --steps 100 --cpu-seconds 0 > gpurun_out/c4ref/c4_h16.json
echo bench done
