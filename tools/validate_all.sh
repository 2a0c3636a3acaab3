#!/bin/bash
# One command to re-establish everything this repository claims.
#   here (no GPU):     tools/validate_all.sh cpu      build (hipcc cross-compiles gfx950) + CPU tests
#   on an MI355X box:  tools/validate_all.sh gpu      GPU parity tests + smoke + default bench line
#                      tools/validate_all.sh profile <tag>   rocprofv3 kernel stats + PMC passes -> gpurun_out/prof_<tag>/
#                                                    (then, back in the build container: python tools/summarize_round.py <tag>)
# A failing step stops the script (no further GPU step is started after a failure).
set -euo pipefail
cd "$(dirname "$0")/.."
case "${1:-cpu}" in
  cpu)
    python __graft_entry__.py
    python -m pytest tests -x -q -m "not gpu"
    ;;
  gpu)
    timeout -k 10 900 python -m pytest tests -x -q -m gpu
    python -c "import __graft_entry__ as g; g.smoke()"
    python bench.py
    ;;
  profile)
    tools/profile_round.sh "${2:?tag}"
    ;;
  *)
    echo "usage: $0 cpu|gpu|profile <tag>"; exit 2;;
esac
