cd $GRAFT_REPO_ROOT
B="bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-dp-path --train-loop-steps 0 --no-val-auroc"
echo "one job:"; python3 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "two jobs at once:"
python3 $B > /tmp/j1.txt 2>/dev/null &
P1=$!
python3 $B > /tmp/j2.txt 2>/dev/null &
P2=$!
wait $P1 $P2
for f in /tmp/j1.txt /tmp/j2.txt; do tail -1 $f | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
