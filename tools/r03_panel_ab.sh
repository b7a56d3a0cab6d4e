# weight-panel cache against per-call packs on one device: default bench line, two rounds each
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_panel_cache_ab.txt
echo "# bench.py --no-cpu-baseline --no-extras (22 clips, 20 timed steps); B side: ZSV_NO_PANEL_CACHE=1" > $O
for rep in 1 2; do
  for v in cache nocache; do
    if [ $v == nocache ]; then export ZSV_NO_PANEL_CACHE=1; else unset ZSV_NO_PANEL_CACHE; fi
    python3 $R/bench.py --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.readline())
print('$v rep $rep: %.1f clips/s  %.3f ms/step  forward %.2f ms  forward+backward %.2f ms  host enqueue (idle queue) %.1f ms' % (
    j['value'], j['ms_per_step'], j['phases']['forward_ms'], j['phases']['forward_backward_ms'], j['host_enqueue_ms']['idle_queue']))" >> $O
  done
done
unset ZSV_NO_PANEL_CACHE
cat $O
