cd /root/repo
for shape in "80,240,30" "100,300,40" "110,330,35" "72,200,25"; do
  echo "# shape $shape"
  for cfg in "0 0" "4 256" "4 512" "8 256"; do
    set -- $cfg
    envs=""
    [ "$1" != 0 ] && envs="DAQP_AMD_WG_WAVES=$1 DAQP_AMD_WG_GRID=$2"
    echo -n "waves ${1/#0/default}, in flight ${2/#0/default}: "
    env C4_SHAPE=$shape $envs timeout 600 python tools/c4_rate.py 4096 2>/dev/null | tail -1
  done
done
