# sustained rate: does a long stream of pairs slow down, and is it the socket (power / clock / temperature) or the host?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py --steps 60 --warmup 5 --no-alt --no-latency --no-cpu-baseline --no-roofline > /tmp/b.json 2>/dev/null &
PID=$!
for i in $(seq 1 32); do
  sleep 10
  echo "t=$((i*10))s $(rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "Socket Graphics\|sclk\|junction\|hotspot" | sed 's/GPU\[0\]\t\t: //' | tr '\n' '|' | cut -c1-220)  host rss $(ps -o rss= -p $PID | tr -d ' ') kB"
  kill -0 $PID 2>/dev/null || break
done > gpurun_out/r06/soak_watch.txt 2>&1
wait $PID
python -c "import json; d=json.load(open('/tmp/b.json')); print('bench', d['value'], d['ms_per_step'], d['host_cores_busy'])" >> gpurun_out/r06/soak_watch.txt
cat gpurun_out/r06/soak_watch.txt
