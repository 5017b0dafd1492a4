# Pin the floating-point path (U-Net logits, BatchNorm, tfa AdamW, loss scaling) to the REAL reference.
#
# Neither the build container nor the GPU box can run TensorFlow (no wheel, no network), so this is a MAINTAINER action on any
# machine with Python 3.8-3.10 and network access -- a CPU is enough, the reference's hot path is TF/Keras on whatever device:
#
#   make -f tools/keras_goldens.mk keras-goldens REFERENCE=/path/to/InconsistencyMasks
#   git add tests/golden/keras_*.npz && git commit -m "tests: Keras goldens (U-Net parity pinned)"
#
# Versions: what the reference implies -- tf.keras.mixed_precision + tensorflow_addons' AdamW +
# Keras 2 `model.fit`: TensorFlow 2.10-2.12, tensorflow_addons <= 0.20 (0.20.0 is the last release, built for TF 2.10-2.12),
# numpy < 1.24 (TF 2.10 / 2.11 still use np.object aliases).  The fixtures hold arrays only (weights, inputs, per-layer activations,
# probabilities, losses, post-fit weights) plus a JSON string of the versions: nothing of the reference's source travels.
# With the files present, `python -m pytest tests/test_gpu_keras_goldens.py -m gpu` checks the HIP path AND oracle/unet_oracle.py
# against them (probabilities, every stored layer, 3 optimizer steps); DESIGN.md section 6 then drops "parity unpinned" for the U-Net.
REFERENCE ?= /root/reference
VENV ?= .venv-keras-goldens
PY ?= python3

keras-goldens:
	$(PY) -m venv $(VENV)
	$(VENV)/bin/pip install "tensorflow==2.10.1" "tensorflow_addons==0.20.0" "numpy<1.24" "protobuf<3.20"
	$(VENV)/bin/python tools/dump_keras_goldens.py --reference $(REFERENCE) --out tests/golden --steps 3
	@ls -la tests/golden/keras_*.npz

.PHONY: keras-goldens
