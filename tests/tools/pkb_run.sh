for v in full nomatrix nofilter nomf nearest nearest_nomatrix; do for geo in "0 1" "0 2" "1 2" "2 4"; do timeout 60 tests/tools/pkb_$v.bin "$v wx=$geo" $geo; done; done
