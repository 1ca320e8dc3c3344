#!/bin/bash
# kernel-trace averages of the frame-loop scenes at 1080p (which kernels a frame of MotionBlur / Life / Multipass spends its time in)
export TMPDIR=/tmp
for scene in MotionBlur Life Multipass; do
  rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_clock_$scene -o t -- python3 tools/profile_frame_loop.py $scene 2>&1 | grep "frames/s"
  head -7 gpurun_out/prof_clock_$scene/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
