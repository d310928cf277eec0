#!/bin/bash
# A/B of the unordered two-phase schedule on the configs it targets.  usage: scripts/ab_unordered.sh
for Q in count location; do
  python scripts/run_query.py --config c4 --query $Q --opt unordered=0 --stats
  for V in 4 8 16 24 32 48; do python scripts/run_query.py --config c4 --query $Q --opt unordered=1 --opt leaf_vote=$V; done
  python scripts/run_query.py --config c4 --query $Q --opt unordered=1 --stats
done
python scripts/run_query.py --config c3 --query any --opt unordered=0 --steps 10
for V in 8 16 32; do python scripts/run_query.py --config c3 --query any --opt unordered=2 --opt leaf_vote=$V --steps 10; done
python scripts/run_query.py --config c5i --query any --opt unordered=0
python scripts/run_query.py --config c5i --query any --opt unordered=2
python scripts/run_query.py --config c5i --query any --opt unordered=2 --opt steal=0
python scripts/run_query.py --config c2 --query count --opt unordered=0
python scripts/run_query.py --config c2 --query count --opt unordered=1
python scripts/run_query.py --config c3 --query count --opt unordered=0 --steps 10
python scripts/run_query.py --config c3 --query count --opt unordered=1 --steps 10
