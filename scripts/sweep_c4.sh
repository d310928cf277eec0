for Q in count location; do
for O in "" "--opt tile=2" "--opt tile=2 --opt leaf_vote=32" "--opt block_size=256" "--opt tile=2 --opt block_size=256" "--opt unordered=0 --opt tile=2" "--opt unordered=0 --opt steal=64" "--opt unordered=0 --opt steal=64 --opt tile=2" "--opt adaptive=0" "--opt tile=2 --opt xcd_chunk=32"; do
python scripts/run_query.py --config c4 --query $Q $O 2>/dev/null
done; done
