#!/bin/bash
# One GPU-box pass that produces everything kept under profiles/ for a round:
#   tools/profile_round.sh r05 [part]       (run from the repo root on the MI355X box; writes gpurun_out/<tag>/)
# part 1: the default bench line exactly as the driver runs it, kernel trace + stats, the two HBM counter passes (FETCH_SIZE, WRITE_SIZE;
#         separate runs, kernel trace only), one SQ counter pass
# part 2: 1M signatures, 125k words = config 4's per-GPU share (with parity + CPU legs), ORB stream = config 3;  part 3: the config 5 stand-ins (replay, replay_growing);
# part 4: 10^6 words on one GPU (config 4's whole vocabulary) with parity + CPU legs
set -u
TAG=${1:-r03}
PART=${2:-all}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"

kstats() {   # $1 = directory of a rocprofv3 --kernel-trace --stats csv run, $2 = output text
    python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if rows and "AverageNs" not in rows[0]:
    import shutil
    shutil.copy(f[0], sys.argv[2])
    sys.exit(0)
with open(sys.argv[2], "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats summary (csv kernel_stats), durations in ns\n")
    out.write("%-110s %8s %14s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0]
        out.write("%-110s %8s %14s %12.1f %12s %12s %7s\n" % (name[:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
}

if [ "$PART" = all ] || [ "$PART" = 1 ]; then
# 1. the default command, exactly as the driver runs it
( cd $ROOT && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/bench.err )
( cd $ROOT && timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_200steps.json 2> $O/bench200.err )
cd /tmp
# 2. kernel trace + stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt_bench.json 2> $O/kt.err
kstats $O/kt $O/${TAG}_bench_kernel_trace.txt
# the launch timeline of the DRIVER's command (20 steps + the drain): every fused launch of its timed region
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt20 -o kt -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2> $O/kt20.err
python $ROOT/tools/pair_stats.py $(find $O/kt20 -name '*kernel_trace.csv' | head -1) "driver's command" --dump > $O/${TAG}_dispatch_timeline.txt 2>&1
rm -rf $O/kt20
# 3. HBM traffic: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- $BENCH > /dev/null 2> $O/pmc_$c.err
done
# 4. SQ counters (occupancy / wait / MFMA busy)
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/pmc_SQ -o pmc -- $BENCH > /dev/null 2> $O/pmc_SQ.err
cd $ROOT
python tools/make_pmc_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras" > $O/${TAG}_pmc.json 2>> $O/pmc_FETCH_SIZE.err
python tools/pmc_summary.py $(dirname $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)) "FETCH_SIZE (KB as rocprofv3 reports it)" > $O/${TAG}_pmc_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py $(dirname $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)) "WRITE_SIZE (KB as rocprofv3 reports it)" > $O/${TAG}_pmc_WRITE_SIZE.txt 2>&1
python tools/pmc_summary.py $(dirname $(find $O/pmc_SQ -name '*counter_collection.csv' | head -1)) "SQ counters" > $O/${TAG}_pmc_SQ.txt 2>&1
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ
fi

if [ "$PART" = all ] || [ "$PART" = 2 ]; then
cd $ROOT
# 10^6 signatures: launch B becomes the dominant kernel; HBM traffic of THIS memory measured by the run itself
timeout 900 python bench.py --signatures 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --pmc > $O/${TAG}_bench_1m.json 2> $O/bench_1m.err
# config 4's per-GPU share: 125 000 words, persistent filter workgroups -- WITH the parity block and the CPU legs; traffic measured by the run
timeout 1200 python bench.py --words 125000 --steps 200 --warmup 20 --no-extras > $O/${TAG}_bench_125k_words.json 2> $O/bench_125k_words.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt125 -o kt -- python $ROOT/bench.py --words 125000 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2> $O/kt125.err
kstats $O/kt125 $O/${TAG}_kernel_trace_125k_words.txt
rm -rf $O/kt125
cd $ROOT
# config 3: ORB stream, 2 000 frames from an empty dictionary; the Hamming scan's average from a kernel trace of the same command
timeout 900 python bench.py --config orb_stream > $O/${TAG}_bench_orb.json 2> $O/bench_orb.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktorb -o kt -- python $ROOT/bench.py --config orb_stream > /dev/null 2> $O/ktorb.err
kstats $O/ktorb $O/${TAG}_kernel_trace_orb.txt
rm -rf $O/ktorb
cd $ROOT
fi

if [ "$PART" = all ] || [ "$PART" = 3 ]; then
cd $ROOT
# config 5 stand-ins: the replay with revisits (fixed world, memory grown to 10^6 signatures) and the one whose dictionary grows from empty
timeout 1500 python bench.py --config replay --signatures 1000000 > $O/${TAG}_bench_replay_1m.json 2> $O/bench_replay.err
timeout 2400 python bench.py --config replay_growing --signatures 1000000 > $O/${TAG}_bench_replay_growing_1m.json 2> $O/bench_replay_growing.err
fi

if [ "$PART" = all ] || [ "$PART" = 4 ]; then
cd $ROOT
# config 4's whole vocabulary on ONE GPU: 10^6 words, with the parity block and the CPU legs (the oracle's exact scan is ~15 s per frame: few steps)
timeout 2400 python bench.py --words 1000000 --steps 8 --warmup 2 --no-extras > $O/${TAG}_bench_1m_words.json 2> $O/bench_1m_words.err
fi
ls -la $O
