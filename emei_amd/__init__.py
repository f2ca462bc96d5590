"""emei_amd — MI355X-native vectorised env-step engine behind emei's EmeiEnv surface."""
__version__ = "0.1.0"
