#!/bin/bash
# Run HERE (no GPU needed) after tools/collect_profiles.sh has run on the GPU box and its output has come back:
# turns gpurun_out/<tag> into the committed summaries profiles/<tag>_bench_kernel_stats.txt,
# profiles/<tag>_bench_pmc_100h.txt and profiles/pmc_traffic.json (hash of the current csrc/).
#   tools/publish_profiles.sh r03_f
set -e
tag=$1
src=gpurun_out/$tag
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
hash=$(python3 -c "from bench import csrc_hash; print(csrc_hash())")
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu   (csrc $hash)"
  python3 tools/rocpd_summary.py stats "$src"/stats/*.db
} > profiles/${tag}_bench_kernel_stats.txt
{
  echo "# rocprofv3 --pmc <one pass per line of tools/collect_profiles.sh> -- python3 bench.py --steps 3 --warmup 1 --no-cpu   (csrc $hash)"
  python3 tools/pmc_summary.py --json profiles/pmc_traffic.json --units 12000000 --fetch-scale 2 \
      "$src"/pmc_fetch "$src"/pmc_write "$src"/pmc_sq "$src"/pmc_sq2
} > profiles/${tag}_bench_pmc_100h.txt
for f in bench_default bench_forcedist; do
  [ -f "$src/$f.log" ] && tail -n 1 "$src/$f.log" > profiles/${tag}_$f.log
done
ls -la profiles/${tag}_*
