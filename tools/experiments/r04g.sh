R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
python tools/experiments/probe_ingest.py --batch 512 --reps 8 2>&1 | tail -2
python tools/experiments/probe_ingest.py --batch 256 --reps 8 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r04g
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04g -o p -- python3 $R/tools/experiments/probe_ingest.py --batch 512 --reps 6 > $O/prof_r04g.log 2>&1 || exit 1
cp $(find $O/prof_r04g -name '*kernel_stats.csv' | head -1) $O/r04g_jpeg_kernel_stats.csv
head -24 $O/r04g_jpeg_kernel_stats.csv | cut -d'(' -f1,3 | cut -c1-200 | awk -F, '{print $1, $(NF-5), $(NF-4), $(NF-3)}'
