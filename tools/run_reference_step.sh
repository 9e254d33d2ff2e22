#!/bin/bash
# Builder-side gpurun wrapper of tools/run_reference_step.py.  Run in the BUILD container:
#   bash tools/run_reference_step.sh stage      # read-only copy of /root/reference/rave -> oracle/_ref/reference (git-ignored)
#   gpurun --timeout 600 -- 'python tools/run_reference_step.py > gpurun_out/reference_step.log 2>&1'
#   bash tools/run_reference_step.sh unstage    # remove the copy again (reference sources never enter the history)
set -e
cd "$(dirname "$0")/.."
case "$1" in
  stage)   mkdir -p oracle/_ref/reference && cp -r /root/reference/rave oracle/_ref/reference/ && find oracle/_ref/reference -name __pycache__ -prune -exec rm -rf {} + && chmod -R a-w oracle/_ref/reference && echo staged ;;
  unstage) chmod -R u+w oracle/_ref/reference 2>/dev/null || true; rm -rf oracle/_ref/reference && echo removed ;;
  *) echo "usage: $0 stage|unstage"; exit 2 ;;
esac
