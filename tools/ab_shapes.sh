B=$1
for n in 4096,4096 1024,1024 256,250,200 128,512,520 512; do
  timeout 100 python tools/time_euler2.py $n 300 2>&1 | tail -1 | sed 's/^/A /'
  PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B timeout 100 python tools/time_euler2.py $n 300 2>&1 | tail -1 | sed 's/^/B /'
done
