for cfg in "4,14,4,7,16,14,16,14" "4,14,4,7,16,14,8,14" "4,14,4,7,8,14,16,14" "2,14,4,7,16,14,16,14" "4,14,4,15,16,14,16,14" "4,14,3,10,16,14,16,14" "4,10,4,7,16,14,16,10" "3,14,4,7,12,14,12,14"; do
  echo "== $cfg"; BSX_SEG_TILES=$cfg python bench.py --no-extra-configs --no-cpu-baseline --steps 10 --dump-launches /tmp/l.txt > /tmp/b.json 2>/dev/null; grep -E "seg_|frame_program" /tmp/l.txt | awk '{printf "%s %s  ", $2, $3}'; python -c "import json; d=json.load(open('/tmp/b.json')); print(' fps', d['value'])"
done
