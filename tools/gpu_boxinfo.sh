#!/bin/bash
echo "nproc: $(nproc)  nproc --all: $(nproc --all)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"
echo "v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)"
cat /proc/self/cgroup | head -5
echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>&1)"
grep -c ^processor /proc/cpuinfo; grep "model name" /proc/cpuinfo | head -1
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
free -g | head -2
env | grep -i -E "omp|thread|cpu" | head
