for v in "" ru16 ru32; do
  if [ -n "$v" ]; then export VQHIP_SO=$PWD/tools/variants/libvqhip_$v.so; else unset VQHIP_SO; fi
  echo "== variant '$v'"; python tools/time_stage.py 256 2>&1 | grep "C=1024"; python tools/time_stage.py 128 2>&1 | grep "C=4096"
done
