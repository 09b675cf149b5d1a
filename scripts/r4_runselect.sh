#!/bin/bash
# round 4, VERDICT item 7: the run select of the split kernel on maps of mixed widths (MIX=1: 1..4 bins a target; MIX=2: some of 5..200 as
# well), A/B with checksums; fuzz; the suite's split tests
for mix in 1 2; do
for v in "" "GANON_HIP_NO_RUN_SELECT=1"; do
  echo "== MIX=$mix, pre-pass: $v"; env MIX=$mix $v python scripts/split_lowcut.py 4 2>&1 | tail -1
  echo "== MIX=$mix, no pre-pass: $v"; env MIX=$mix NO_PREPASS=1 $v python scripts/split_lowcut.py 3 2>&1 | tail -1
  echo "== MIX=$mix, cutoff 0.75: $v"; env MIX=$mix NO_PREPASS=1 $v python scripts/split_lowcut.py 3 2000000 0.75 2>&1 | tail -1
  echo "== MIX=$mix, cutoff 0.75, pre-pass: $v"; env MIX=$mix $v python scripts/split_lowcut.py 3 2000000 0.75 2>&1 | tail -1
done
done
echo "== uniform 2"; python scripts/split_lowcut.py 3 2>&1 | tail -1
for seed in 1 2 3; do SEED=$seed N_CFG=24 timeout 900 python scripts/fuzz_split.py 2>&1 | grep -v "same True oracle True"; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "run_select or packed_select or candidate_select or split" 2>&1 | tail -5
