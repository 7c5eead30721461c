# Every BASELINE.json config + the variants of the bench line, 5 timed steps each (one MI355X): gpurun -- 'bash tools/run_configs.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
short() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; h=r.get('hbm') or {}; f=r.get('fourier') or {}
        print('   %.1f img/s  %.2f ms/step  conv %.0f TFLOP/s = %.3f of peak (%.2f ms)  norm %.2f ms at %.0f GB/s  %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['conv_ms_per_step'], h.get('ms',0), h.get('achieved',0), ('fourier %.2f ms at %.0f GB/s = %.3f' % (f['ms'], f['achieved'], f['frac'])) if f else ''))
"; }
echo "== [2] ResNet-101 MRFP+ 16x768^2 bf16 (the bench line)"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
echo "== [2] + multi-resolution Fourier amplitude mix"; python bench.py --fourier --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_bench_fourier.json | short
echo "== [2] float16"; python bench.py --dtype f16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
echo "== [2] hipGraph replay"; python bench.py --graph --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
echo "== [1] ResNet-50 MRFP+ 8x512^2 fp32"; python bench.py --trunk resnet-50 --size 512 --batch 8 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
echo "== [1] bf16"; python bench.py --trunk resnet-50 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
echo "== [1] bf16 hipGraph"; python bench.py --trunk resnet-50 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --graph 2>/dev/null | short
echo "== [4] WiderResNet-38 MRFP+ 2x1024x2048 bf16"; python bench.py --trunk wider_resnet38_a2 --size 1024 --width 2048 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | short
