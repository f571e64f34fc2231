export PYTHONUNBUFFERED=1 DBG_SLAB_TRIALS=1
for rep in 1 2; do STEPS=6 bash tools/ab_libs.sh main hfold h1mul; done
