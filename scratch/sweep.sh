set -e
for r in 8 32 64; do
  sed "s/#define MI_PLOC_RADIUS 16/#define MI_PLOC_RADIUS $r/" pathtracer_amd/csrc/mi_bvh_build.h > /tmp/mi_bvh_build.h
  echo "== radius $r"
done
