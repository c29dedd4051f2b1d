"""Import harness for the REFERENCE Python (self_play.py / model.py / GoEnv) — fixture generation only.

Runs ONLY in the build container, where /root/reference exists.  Nothing in tests/, bench.py or the product
imports this module at run time on the GPU box: the committed fixtures under tests/golden/ are what travels.

What it does: the reference loads "./GoEnv/go_env.so" relative to the cwd
(GoEnv/environment.py:42) and decorates classes with @ray.remote (self_play.py:881); ray is not installed.
So we (1) chdir into oracle/_ref (where `make -C oracle ref` put the compiled reference engine),
(2) register a 3-line stub `ray` whose `remote` is the identity, (3) put /root/reference on sys.path.
"""
import os
import sys
import types

REF = os.environ.get("TRANSGO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.path.join(REPO, "oracle", "_ref")


def load_reference():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: fixtures can only be regenerated in the build container")
    if not os.path.exists(os.path.join(REF_BUILD, "GoEnv", "go_env.so")):
        raise RuntimeError("run `make -C oracle ref` first")
    sys.dont_write_bytecode = True
    ray = types.ModuleType("ray")
    ray.remote = lambda c: c
    ray.get = lambda x: x
    sys.modules.setdefault("ray", ray)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF_BUILD)
    import torch  # noqa: F401
    import configure
    import self_play
    import model
    from GoEnv import environment
    cfg = configure.Config()
    cfg.device = torch.device("cpu")
    return types.SimpleNamespace(configure=configure, self_play=self_play, model=model,
                                 environment=environment, Config=configure.Config, cfg=cfg)
