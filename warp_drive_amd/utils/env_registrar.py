"""Registry of environment classes by (name, backend)
(reference warp_drive/utils/env_registrar.py:4-132; the source-path bookkeeping for
nvcc templating has no HIP counterpart because the code object is prebuilt)."""


class EnvironmentRegistrar:
    _backends = ("cpu", "hip")

    def __init__(self):
        self._envs = {}

    def add(self, env_backend="cpu", cuda_env_src_path=None):
        if env_backend in ("pycuda", "numba"):
            env_backend = "hip"
        assert env_backend in self._backends

        def register(cls):
            name = getattr(cls, "name", None)
            assert name, "the environment class needs a `name` attribute"
            key = (name, env_backend)
            if key in self._envs:
                raise Exception(f"{name} for backend {env_backend} is already registered")
            self._envs[key] = cls
            return cls

        return register

    def get(self, name, env_backend="cpu"):
        if env_backend in ("pycuda", "numba"):
            env_backend = "hip"
        return self._envs[(name, env_backend)]

    def has_env(self, name, env_backend="cpu"):
        if env_backend in ("pycuda", "numba"):
            env_backend = "hip"
        return (name, env_backend) in self._envs
