set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ref
timeout -k 5 200 python bench.py > gpurun_out/ref/bench.json 2> gpurun_out/ref/bench.err
echo bench done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ref/ks -- python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-split > gpurun_out/ref/ks.log 2>&1
echo stats done
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ref/fetch -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > gpurun_out/ref/fetch.log 2>&1
echo fetch done
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ref/write -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > gpurun_out/ref/write.log 2>&1
echo write done
