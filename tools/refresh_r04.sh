#!/bin/bash
# Round-4 evidence for profiles/ in two parts (each under the gpurun limit):  bash tools/refresh_r04.sh 1|2  -> gpurun_out/r04/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
if [ "$1" = "1" ]; then
    # C2: bench line, per-kernel stats, HBM traffic (separate FETCH / WRITE passes), SQ counters (two passes) -- all on the rotating inputs
    timeout -k 5 200 python bench.py > $O/bench.json 2> $O/bench.err; echo bench done
    timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-split > $O/ks.log 2>&1; echo stats done
    KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > $O/fetch.log 2>&1; echo fetch done
    KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > $O/write.log 2>&1; echo write done
    rm -rf gpurun_out/sq; bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; echo sq done
    f=$(find $O/ks -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; python3 profiles/summarize.py stats $f > $O/kernel_stats.txt
    python3 profiles/summarize.py traffic $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
    python3 tools/pmc_sq_json.py gpurun_out/sq $O/pmc_sq.json
    # C4 per-kernel stats at 8 and 16 heads
    KM_BENCH_OPTIONS=no_core_merge=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4h8 -o c4 -- python3 bench.py --workload c4 --steps 50 --cpu-seconds 0 > $O/c4_h8_prof.json 2> $O/c4h8.err
    KM_BENCH_OPTIONS=no_core_merge=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4h16 -o c4 -- python3 bench.py --workload c4 --heads 16 --steps 50 --cpu-seconds 0 > $O/c4_h16_prof.json 2> $O/c4h16.err
    { echo "== C4, 8 heads (rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c4 --steps 50)"; python3 profiles/summarize.py stats $(find $O/c4h8 -name "*kernel_stats.csv" | head -1);
      echo "== C4, 16 heads"; python3 profiles/summarize.py stats $(find $O/c4h16 -name "*kernel_stats.csv" | head -1); } > $O/c4_kernel_stats.txt
    echo c4 done
else
    bash tools/refresh_workloads.sh > $O/workloads.log 2>&1; echo workloads done
    bash tools/trace_train.sh 8 > /dev/null; cp gpurun_out/trace_train_8.txt $O/train_trace_8.txt
    bash tools/trace_train.sh 64 > /dev/null; cp gpurun_out/trace_train_64.txt $O/train_trace_64.txt; echo traces done
    bash tools/prof_legacy.sh > /dev/null 2>&1; cp gpurun_out/legacy/kernel_stats.txt $O/legacy_kernel_stats.txt; cp gpurun_out/legacy/bench_prof.json $O/legacy_bench.json; echo legacy done
    { cd tools/micro; ./bin/persist_bench; ./bin/persist_bench --work 2000; ./bin/persist_bench --wgs 512; ./bin/persist_bench --wgs 1024; cd ../..; } > $O/phase_boundary_microbench.txt 2>&1
    { cd tools/micro; for v in narrow2 narrow4 narrow8 wide4 dma; do ./bin/tile_bench --variant $v; done; ./bin/tile_bench --variant narrow2 --bm 64; ./bin/tile_bench --variant dma --bm 64; ./bin/tile_bench --variant dma --k 1024; ./bin/tile_bench --variant dma --k 32; ./bin/dma_probe; ./bin/clock_probe; cd ../..; } > $O/tile_microbench.txt 2>&1
    { ./tools/micro/bin/attn_train_bench 1; ./tools/micro/bin/attn_train_bench 0; } > $O/attn_block_microbench.txt 2>&1
    echo micro done
fi
