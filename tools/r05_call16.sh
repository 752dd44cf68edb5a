python tools/h3_small_ab.py 8,21,24,25,26,28 12608 2>/dev/null
python tools/h3_small_ab.py 8,21,25,26,28 6304 2>/dev/null
