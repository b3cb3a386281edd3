# poll rocm-smi power / clocks while a workload loops (developer aid): usage power_poll.sh <tag> <cmd...>
TAG=$1; shift
( "$@" > /tmp/pp_$TAG.log 2>&1 ) &
PID=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' | sed "s/^/$TAG: /"; echo
  sleep 0.5
done
wait $PID
