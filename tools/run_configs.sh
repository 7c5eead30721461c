cd $GRAFT_REPO_ROOT
echo "== c1 f32"; python bench.py --trunk resnet-50 --size 512 --batch 8 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
echo "== c1 bf16"; python bench.py --trunk resnet-50 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
echo "== c1 bf16 graph"; python bench.py --trunk resnet-50 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --graph 2>&1 | tail -1 | cut -c1-900
echo "== c2 f16"; python bench.py --dtype f16 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
echo "== c2 graph"; python bench.py --graph --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
echo "== c4 wrn"; python bench.py --trunk wider_resnet38_a2 --size 1024 --width 2048 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
