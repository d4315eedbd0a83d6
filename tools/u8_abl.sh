python tools/first_u8_bench.py > gpurun_out/first_u8_bench2.txt 2>&1
for a in 0 1 2 4 8 15; do echo "== ABL $a" >> gpurun_out/first_u8_bench2.txt; AUGCL_U8_ABL=$a python tools/first_u8_bench.py tuning 2>&1 | grep "first layer" >> gpurun_out/first_u8_bench2.txt; done
cat gpurun_out/first_u8_bench2.txt
