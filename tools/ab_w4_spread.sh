P=multilingual-image-captioning_amd
cp $P/libmic_hip.so /tmp/lib_new.so
for arm in new s1 s2 new s1 s2; do
  if [ $arm = new ]; then cp /tmp/lib_new.so $P/libmic_hip.so; else cp $P/libmic_hip_$arm.so $P/libmic_hip.so; fi
  echo "== $arm"; python tools/bench_head_fwd.py 2>&1 | grep -v "amdgpu.ids\|MIC_GEMM"
done
cp /tmp/lib_new.so $P/libmic_hip.so
