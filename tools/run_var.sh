# usage: tools/run_var.sh tag1 tag2 ...   ("" = the in-tree library).  CHECK=1: also screened-vs-exact agreement (tools/screen_check.py --quick)
for v in "" "$@"; do
  if [ -n "$v" ]; then export VQHIP_SO=$PWD/tools/variants/libvqhip_$v.so; else unset VQHIP_SO; fi
  echo "== variant '$v'"; python tools/time_assign.py 2>&1 | tail -2
  if [ -n "$CHECK" ] && [ -n "$v" ]; then python tools/screen_check.py --quick 2>&1 | grep -c "idx_equal=True (bad 0) q_equal=True"; python tools/screen_check.py --quick 2>&1 | grep -v "idx_equal=True (bad 0) q_equal=True" | tail -3; fi
done
