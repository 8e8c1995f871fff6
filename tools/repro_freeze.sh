#!/bin/bash
# Runs tools/repro_freeze.py <mode> <loops> under a watchdog: when the log has not grown for STALL seconds the process is
# taken for frozen -- rocgdb attaches and dumps every thread's native stack, then the exact PID is killed.
# usage: tools/repro_freeze.sh <mode> <loops> <outdir> [STALL=240] [LIMIT=1500]
mode=$1; loops=$2; out=$3; STALL=${4:-240}; LIMIT=${5:-1500}
mkdir -p "$out"
SG_REPRO_FH="$out/$mode.fh.txt" python tools/repro_freeze.py "$mode" "$loops" > "$out/$mode.log" 2>&1 &
pid=$!
start=$(date +%s); last_size=-1; last_change=$start
while kill -0 $pid 2>/dev/null; do
  sleep 5
  now=$(date +%s); size=$(stat -c %s "$out/$mode.log")
  if [ "$size" != "$last_size" ]; then last_size=$size; last_change=$now; fi
  if [ $((now - last_change)) -ge $STALL ] || [ $((now - start)) -ge $LIMIT ]; then
    echo "[watchdog] no progress for $((now - last_change)) s (elapsed $((now - start)) s): dumping stacks of $pid" >> "$out/$mode.log"
    rocm-smi --showuse --showmemuse > "$out/$mode.smi.txt" 2>&1
    for i in 1 2 3; do kill -USR1 $pid; sleep 2; done            # three samples of every thread's Python stack
    cat /proc/$pid/status > "$out/$mode.status.txt" 2>&1; ls /proc/$pid/task | wc -l >> "$out/$mode.status.txt"
    for t in /proc/$pid/task/*; do echo "$t $(cat $t/comm) $(cat $t/wchan 2>/dev/null) $(awk '{print $3, $14, $15}' $t/stat)"; done > "$out/$mode.threads.txt" 2>&1
    kill -9 $pid
    echo "[watchdog] killed $pid" >> "$out/$mode.log"
    break
  fi
done
wait $pid 2>/dev/null
echo "[watchdog] exit code $?" >> "$out/$mode.log"
tail -25 "$out/$mode.log"
