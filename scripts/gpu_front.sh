# front-kernel variants: time layer 1 of the timed forward
for cfg in "8 1" "4 2" "4 3" "8 2"; do
  set -- $cfg
  echo "== TH=$1 blocks/CU=$2"
  YOLO_FRONT_TH=$1 YOLO_FRONT_BPC=$2 timeout -k 10 200 python -u bench.py --steps 3 --warmup 2 --no-cpu-baseline --layer-table gpurun_out/lt.txt > /dev/null 2>&1 && sed -n '3p' gpurun_out/lt.txt
done
