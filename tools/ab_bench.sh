#!/bin/bash
# Same-box A/B of two builds of libsrhip.so (box-to-box variance of the bench is +-2.5 %, run-to-run on one box +-0.3 %).
# In the build container:   tools/ab_bench.sh prepare      -> libsrhip_A.so = HEAD (git stash), libsrhip_B.so = working tree
# On the GPU box (gpurun):  tools/ab_bench.sh run [bench args]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "prepare" ]; then
  if git diff --quiet && git diff --cached --quiet; then echo "working tree equals HEAD: nothing to compare"; exit 1; fi
  git stash -q && python -c "import semireward_amd.build as b; b.build()" > /dev/null && cp semireward_amd/libsrhip.so semireward_amd/libsrhip_A.so
  git stash pop -q && python -c "import semireward_amd.build as b; b.build()" > /dev/null && cp semireward_amd/libsrhip.so semireward_amd/libsrhip_B.so
  echo "prepared A (HEAD) and B (working tree); remove semireward_amd/libsrhip_[AB].so afterwards"
else
  shift || true
  for i in 1 2 3; do for v in A B; do
    cp semireward_amd/libsrhip_$v.so semireward_amd/libsrhip.so
    echo -n "$v "; python bench.py --no-cpu-baseline --no-roofline "$@" | grep -o '"value": [0-9.]*'
  done; done
  cp semireward_amd/libsrhip_B.so semireward_amd/libsrhip.so
fi
