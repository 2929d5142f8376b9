for N in 12 13 14 15 16 18 20; do for w in 1 2; do
 r=$(BROV_DEV_FUSED_WAVES=$w python bench.py --config 5 --horizon $N --no-cpu-baseline --batch 16384 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print(round(o['value']/1e6,2))")
 echo "N=$N waves=$w  $r M"
done; done
