cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06y}_campaign; mkdir -p $O
DAQP_CAMPAIGN_SALT=${2:-6} DAQP_AMD_IMG_MIN_BATCH=1 DAQP_AMD_WG_TIER_MIN_BATCH=1 timeout 1500 python tools/parity_campaign.py 30000 30000 300 60 > $O/parity_campaign_img_tier.txt 2>&1
DAQP_AMD_WG_TIER_MIN_BATCH=1 timeout 900 python tools/large_shapes.py > $O/large_shapes_tier.txt 2>&1
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 timeout 900 python tools/full_size_parity.py C2,C3,C4,C5 1 > $O/full_size_parity_default.txt 2>&1
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 timeout 900 python tools/full_size_parity.py C2,C3,C4,C5 1 exact > $O/full_size_parity_exact.txt 2>&1
for f in parity_campaign_img_tier large_shapes_tier; do tail -n 2 $O/$f.txt | cut -c1-200; done
grep -h "^C[2345]:" $O/full_size_parity_*.txt | cut -c1-220
