set -e
run() { echo "== $*"; timeout -k 10 280 env "$@"; }
run CUDDLE_LEAF_GROUPED=1 python -u tools/random_parity_sweep.py --cases 50 --seed 301
run CUDDLE_LEAF_GROUPED=1 python -u tools/random_parity_sweep.py --cases 40 --seed 302 --trees
run CUDDLE_LEAF_GROUPED=1 python -u tools/random_parity_sweep.py --cases 30 --seed 303 --sharded
run CUDDLE_LANES=2 python -u tools/random_parity_sweep.py --cases 50 --seed 304
run CUDDLE_LANES=2 CUDDLE_LEAF_GROUPED=1 python -u tools/random_parity_sweep.py --cases 40 --seed 305 --boundary
run CUDDLE_RABITQ_GROUPED=1 python -u tools/random_parity_sweep.py --cases 40 --seed 306
run CUDDLE_X=0 python -u tools/random_parity_sweep.py --cases 50 --seed 307
run CUDDLE_X=0 python -u tools/random_parity_sweep.py --cases 40 --seed 308 --trees
run CUDDLE_X=0 python -u tools/random_parity_sweep.py --cases 30 --seed 309 --sharded
run CUDDLE_X=0 python -u tools/random_parity_sweep.py --cases 30 --seed 310 --boundary
run CUDDLE_LDS_DIST_BYTES=700 python -u tools/random_parity_sweep.py --cases 40 --seed 311
run CUDDLE_LDS_DIST_BYTES=700 python -u tools/random_parity_sweep.py --cases 30 --seed 312 --trees
run CUDDLE_SCAN_ORDER=1 python -u tools/random_parity_sweep.py --cases 40 --seed 313
run CUDDLE_SCAN_ORDER=1 python -u tools/random_parity_sweep.py --cases 30 --seed 314 --sharded
run CUDDLE_REFINE_BUCKETS=1 python -u tools/random_parity_sweep.py --cases 40 --seed 315
run CUDDLE_REFINE_BUCKETS=2 python -u tools/random_parity_sweep.py --cases 30 --seed 316 --trees
