#!/bin/bash
# Runs tools/repro_freeze.py <mode> <loops> under a watchdog: when the log has not grown for STALL seconds the process is
# taken for frozen -- rocgdb attaches and dumps every thread's native stack, then the exact PID is killed.
# usage: tools/repro_freeze.sh <mode> <loops> <outdir> [STALL=240] [LIMIT=1500]
mode=$1; loops=$2; out=$3; STALL=${4:-240}; LIMIT=${5:-1500}
mkdir -p "$out"
python tools/repro_freeze.py "$mode" "$loops" > "$out/$mode.log" 2>&1 &
pid=$!
start=$(date +%s); last_size=-1; last_change=$start
while kill -0 $pid 2>/dev/null; do
  sleep 5
  now=$(date +%s); size=$(stat -c %s "$out/$mode.log")
  if [ "$size" != "$last_size" ]; then last_size=$size; last_change=$now; fi
  if [ $((now - last_change)) -ge $STALL ] || [ $((now - start)) -ge $LIMIT ]; then
    echo "[watchdog] no progress for $((now - last_change)) s (elapsed $((now - start)) s): dumping stacks of $pid" >> "$out/$mode.log"
    rocm-smi --showuse --showmemuse > "$out/$mode.smi.txt" 2>&1
    timeout 120 /opt/rocm/bin/rocgdb -p $pid -batch -ex "set pagination off" -ex "thread apply all bt 25" > "$out/$mode.gdb.txt" 2>&1
    kill -9 $pid
    echo "[watchdog] killed $pid" >> "$out/$mode.log"
    break
  fi
done
wait $pid 2>/dev/null
echo "[watchdog] exit code $?" >> "$out/$mode.log"
tail -25 "$out/$mode.log"
