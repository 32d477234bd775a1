import sys
sys.path.insert(0, '.')
from deepcgp_amd import device as dev
ctx = dev.get_context()
for i in range(4):
    print("api mfma TF/s", ctx.measured_mfma_f64_tflops())
print("store", ctx.measured_store_gbs(320, 144, 256))
