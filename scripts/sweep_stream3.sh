python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --flat --opt stream=0 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --flat --opt stream=2 --opt stream_rays=64 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --flat --opt stream=2 --opt stream_rays=256 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --flat --opt stream=2 2>/dev/null
for C in "c3 any" "c3 closest" "c3 count" "c5s closest"; do set -- $C
python scripts/run_query.py --config $1 --query $2 --steps 8 --opt stream=2 2>/dev/null
python scripts/run_query.py --config $1 --query $2 --steps 8 --opt stream=2 --opt stream_refill=16 2>/dev/null
done
python scripts/run_hash.py --n 2097152 --mesh headline --opt stream=2 2>/dev/null
python scripts/run_hash.py --n 1048576 --mesh headline --opt stream=2 --opt stream_rays=256 2>/dev/null
python scripts/run_hash.py --n 1048576 --mesh bunny --opt stream=2 --opt stream_rays=256 2>/dev/null
