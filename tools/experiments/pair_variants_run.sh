#!/bin/bash
# the pair form's duration on one FinalExp commitment for the default library and each build/pair_<name> variant (pair_variants.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo default $(python3 tools/experiments/pair_check.py --timing | tail -1)
for name in "$@"; do
  echo $name $(STARKHIP_LIBRARY=$R/build/pair_$name/libstarkhip_pair.so python3 tools/experiments/pair_check.py --timing | tail -1)
done
