cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in 176 224 240 248 256; do
  echo "cap $cap"
  for i in 1 2; do GDM_DCNN_CAP=$cap python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
done
