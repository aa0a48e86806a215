"""edgestyle_amd — MI355X-native EdgeStyle multi-ControlNet SD1.5 denoising hot path."""
