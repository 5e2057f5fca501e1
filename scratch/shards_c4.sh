for e in "$@"; do envs=$(echo $e | tr ',' ' '); echo "== $e"; env $envs timeout -k 10 900 python scratch/shards.py C4-blob1.3M-4k-256spp 1,2,4,8 || exit 1; done
