#!/bin/bash
# Wall time of the training part of the CLI run of Example/ALL_RESULTS.tsv:19 (Influenza-A, 5 voters x 35 epochs) with 1, 2 and 4 voter lanes.
cd /tmp && export TMPDIR=/tmp
for lanes in 1 2 4 1 4; do
  rm -rf /tmp/cli_lanes && mkdir -p /tmp/cli_lanes && cd /tmp/cli_lanes
  s=$(date +%s%N)
  IDELUCS_VOTER_LANES=$lanes PYTHONPATH=$GRAFT_REPO_ROOT python3 -m idelucs_amd --sequence_file $GRAFT_REPO_ROOT/tests/data/Influenza-A.fas --GT_file $GRAFT_REPO_ROOT/tests/data/Influenza-A_GT.tsv \
      --n_clusters 5 --n_epochs 35 --n_voters 5 --batch_sz 512 --k 6 > log.txt 2>&1
  e=$(date +%s%N)
  echo "lanes=$lanes wall $(( (e - s) / 1000000 )) ms; $(grep -a -o 'ACC: [0-9.]*' log.txt | tail -1)"
done
