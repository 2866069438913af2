exec < /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /tmp/counters.txt 2>&1; wc -l /tmp/counters.txt; head -30 /tmp/counters.txt | cut -c1-200; grep -o "\bTC[PC]_[A-Z0-9_]*\|\bTA_[A-Z0-9_]*\|\bSQ_[A-Z0-9_]*" /tmp/counters.txt | sort -u | tr '\n' ' ' | fold -w 250
