#!/bin/bash
# usage: clk_probe.sh <label> <cmd...>
label=$1; shift
"$@" > /tmp/out_$label.txt 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 1; done
wait $pid
tail -1 /tmp/out_$label.txt | cut -c1-200
