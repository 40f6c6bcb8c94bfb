cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o c -- env -C /root/repo python tools/cfgprof.py cfg4_lognormal_full_mantissa 1e9 1 > /dev/null 2>&1
db=$(find /tmp/prof_tl -name '*_results.db' | head -1)
python /root/repo/tools/rocpd_timeline.py $db 60 | cut -c1-120
