#!/bin/bash
# Every profile the bench line and DESIGN.md quote, from ONE binary on ONE box:
#   tools/collect_profiles.sh <tag>      (on the GPU box, from the repo root; e.g. tag = r03)
# Outputs go to profiles/<round>/ (round = the first three characters of the tag):
# 1. rocprofv3 --kernel-trace --stats of the headline run              -> <tag>_bench_n20.{md,csv} (+ idle time between kernels)
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes          -> <tag>_pmc_hbm.md, profiles/pmc_msm_accumulate.json
# 3. --pmc SQ issue / stall set                                        -> <tag>_pmc_sq.md, profiles/pmc_ntt.json
# 4. --pmc GRBM_GUI_ACTIVE (effective clock of the long kernels)       -> <tag>_pmc_clock.md
# Counter passes are their own runs with no trace domain (gpurun refuses --pmc together with tracing); the program after `--` is
# python3 itself (no env / bash hop: the profiler's library has initialised the GPU by then).
set -u
tag=$1
export ZK_ACC_LAUNCHES_PER_PROOF=${ZK_ACC_LAUNCHES_PER_PROOF:-5}
out=gpurun_out/prof_$tag
dir=profiles/${tag:0:3}
mkdir -p "$out" "$dir"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="bench.py --steps 3 --warmup 1 --extra-legs off --streams-leg 0 --no-cpu-baseline"
PMCB="bench.py --steps 1 --warmup 1 --extra-legs off --streams-leg 0 --no-cpu-baseline --no-check"
commit=$(cat .git_head 2>/dev/null || echo unknown)
# a discarded run first: the first minute of a session runs at other clocks than the rest (a traced run at the very start read 3.5 % above
# the bench line that followed it)
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --extra-legs off --streams-leg 0 --no-cpu-baseline > "$out/warmup.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 $BENCH > "$out/trace.log" 2>&1 || exit 1
python3 tools/summarize_rocprof.py "$out/trace" "$dir/${tag}_bench_n20" "rocprofv3 --kernel-trace --stats -- python3 $BENCH" > "$out/trace_summary.txt"
python3 tools/trace_idle.py "$out/trace" >> "$dir/${tag}_bench_n20.md"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o p -- python3 $PMCB > "$out/pmc_$c.log" 2>&1 || exit 1
done
python3 tools/summarize_pmc.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$dir/${tag}_pmc_hbm" "python3 $PMCB" > "$out/pmc_summary.txt"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d "$out/pmc_sq" -o p -- python3 $PMCB > "$out/pmc_sq.log" 2>&1 || exit 1
python3 tools/summarize_sq.py "$out/pmc_sq" "$dir/${tag}_pmc_sq.md" "python3 $PMCB" profiles/pmc_ntt.json "$out/trace" > "$out/sq_summary.txt"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_clk" -o p -- python3 $PMCB > "$out/pmc_clk.log" 2>&1
python3 tools/summarize_clock.py "$out/pmc_clk" "$dir/${tag}_pmc_clock.md" > "$out/clk_summary.txt" 2>&1
tail -n 30 "$out/trace_summary.txt"; cat "$out/pmc_summary.txt" | head -12; head -8 "$out/sq_summary.txt"; cat "$out/clk_summary.txt" | head -12
# profiles/ does not travel back from the GPU box; gpurun_out/ does
mkdir -p "$out/profiles_out" && cp "$dir/${tag}"_bench_n20.md "$dir/${tag}"_bench_n20.csv "$dir/${tag}"_pmc_hbm.md "$dir/${tag}"_pmc_sq.md "$dir/${tag}"_pmc_clock.md profiles/pmc_msm_accumulate.json profiles/pmc_ntt.json "$out/profiles_out/"
