for R in 2048 4096; do
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=0 2>/dev/null
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=2 2>/dev/null
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=2 --opt xcd_chunk=0 2>/dev/null
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=2 --opt stream_rays=4096 2>/dev/null
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=2 --opt stream_rays=1024 --opt stream_refill=48 2>/dev/null
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 2>/dev/null
done
python scripts/run_query.py --config c4 --res 2048 --query count --steps 8 --flat --opt stream=0 2>/dev/null
python scripts/run_query.py --config c4 --res 2048 --query count --steps 8 --flat --opt stream=2 2>/dev/null
python scripts/run_query.py --config c3 --query closest --steps 8 2>/dev/null
python scripts/run_query.py --config c5s --query closest --steps 8 2>/dev/null
