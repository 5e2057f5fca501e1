for e in "$@"; do envs=$(echo $e | tr ',' ' '); echo "== $e"; env $envs timeout -k 10 200 python scratch/shards.py C3-blob82k-1080p-16spp 1,8 || exit 1; done
