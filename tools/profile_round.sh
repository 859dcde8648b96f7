#!/bin/bash
# One GPU-box pass that produces everything kept under profiles/ for a round:
#   tools/profile_round.sh r02        (run from the repo root on the MI355X box; writes gpurun_out/<tag>/)
# kernel trace + stats, the two HBM counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only), one SQ counter
# pass, the default bench line (CPU baselines + parity block), and the secondary configurations (1M signatures, ORB stream, 2 ranks,
# 125k / 1M words).
set -u
TAG=${1:-r02}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline"

# 1. the default command, exactly as the driver runs it
( cd $ROOT && timeout 900 python bench.py > $O/bench.json 2> $O/bench.err )

cd /tmp
# 2. kernel trace + stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt_bench.json 2> $O/kt.err
# 3. HBM traffic: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- $BENCH > /dev/null 2> $O/pmc_$c.err
done
# 4. SQ counters (occupancy / wait / MFMA busy)
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/pmc_SQ -o pmc -- $BENCH > /dev/null 2> $O/pmc_SQ.err

cd $ROOT
python tools/make_pmc_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "python bench.py --steps 100 --warmup 10 --no-cpu-baseline" > $O/${TAG}_pmc.json 2>> $O/pmc_FETCH_SIZE.err
python tools/pmc_summary.py $(dirname $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)) "FETCH_SIZE (KB as rocprofv3 reports it)" > $O/pmc_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py $(dirname $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)) "WRITE_SIZE (KB as rocprofv3 reports it)" > $O/pmc_WRITE_SIZE.txt 2>&1
python tools/pmc_summary.py $(dirname $(find $O/pmc_SQ -name '*counter_collection.csv' | head -1)) "SQ counters" > $O/pmc_SQ.txt 2>&1
find $O/kt -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
# the raw traces are large: keep the per-kernel summaries only
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ

# 5. secondary configurations
timeout 600 python bench.py --signatures 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_1m.json 2> $O/bench_1m.err
timeout 600 python bench.py --config orb_stream --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_orb.json 2> $O/bench_orb.err
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_gpus2.json 2> $O/bench_gpus2.err
# larger vocabularies (config 4's per-GPU share and the whole of it): the persistent filter launch
timeout 300 python bench.py --words 125000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_125k_words.json 2> $O/bench_125k_words.err
timeout 300 python bench.py --words 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_1m_words.json 2> $O/bench_1m_words.err
timeout 200 python tools/bench_knn_sizes.py > $O/knn_sizes.json 2> $O/knn_sizes.err
ls -la $O
