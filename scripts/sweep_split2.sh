#!/bin/bash
# block splitting x tile shape on the image-shaped configs (pruning queries)
for C in "--config c5i" "--config c5i --res 512" "--config c5i --res 2048 --steps 10" "--config c2" "--config c4" "--config c5i --subdiv 6" "--config c5i --query any" "--config c4 --query any" "--config c2 --query first"; do
  case "$C" in *query*) Q="";; *) Q="--query closest";; esac
  for A in "" "--opt split=4" "--opt tile=2 --opt split=4" "--opt tile=2" "--opt tile_small=2 --opt split=4" "--opt tile_small=1 --opt split=4"; do
    python scripts/run_query.py $C $Q $A 2>&1 | tail -1
  done
done
