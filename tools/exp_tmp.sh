P="timeout 600 python3 tools/bundle_probe.py time 131"
for a in "1000 1 smooth" "1000 1 checker" "500 1 smooth" "250 1 smooth" "125 1 smooth" "60 1 smooth" "375 1 smooth" "700 1 smooth" "1300 1 smooth" "1600 1 smooth" "256 1 rough"; do set -- $a; $P $1 16 $3 $2 | cut -c1-230; done
