# same-box A/B of two builds of libmic_hip.so: the tree's library against multilingual-image-captioning_amd/libmic_hip_base.so
# usage: bash tools/ab_lib.sh '<command printing the figure>'   (runs new, base, new, base)
P=multilingual-image-captioning_amd
cp $P/libmic_hip.so /tmp/lib_new.so
for arm in new base new base; do
  if [ $arm = new ]; then cp /tmp/lib_new.so $P/libmic_hip.so; else cp $P/libmic_hip_base.so $P/libmic_hip.so; fi
  echo "== $arm"; eval "$1"
done
cp /tmp/lib_new.so $P/libmic_hip.so
