# the "other configurations" table of DESIGN.md section 3
run() { timeout -k 10 300 python bench.py "$@" --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], round(d['ms_per_step'],1), round(d['patches_per_s'],1), d.get('step_mfma_frac'))" "$@"; }
run --config CNN:XS --global-batch 256
run --config DRCNN:L --global-batch 64
run --config Unet:L --global-batch 128
run --config SAUnet:L --global-batch 128 --frames 174
run --config SAUSnet:L --global-batch 256
run --config BLUnet:XXL --global-batch 256
run --config PUnet:XL --global-batch 128
