cd /tmp && export TMPDIR=/tmp
for d in 1 2 3; do
GRX_DBG=$d timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$d -o kt -- python3 /root/repo/tools/one_bfs.py 24 2 5 > /root/repo/gpurun_out/kt.log 2>&1
f=$(find /tmp/kt$d -name "*kernel_trace.csv" | head -1)
echo "== dbg $d"; python3 /root/repo/tools/kt_print.py $f | grep -E "FreshToBitmap|LoadBalanced" 
done
