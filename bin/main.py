"""
Drop-in twin of the reference's bin/main.py (its line 12 passes py4cast.lightning.AutoRegressiveLightning to the
Lightning CLI; this one passes the MI355X implementation -- same constructor, same yaml configs):

    python bin/main.py fit --config config/CLI/trainer.yaml --config config/CLI/dataset/titan.yaml \
                           --config config/CLI/model/halfunet.yaml

Needs `lightning` and the reference package `py4cast` (for its CLI class and PlDataModule) on the PYTHONPATH, plus
this repository's root (so that the plugin module py4cast_plugin_mi355x is discovered).
"""

if __name__ == "__main__":
    try:
        from py4cast.cli import Py4castLightningCLI
        from py4cast.lightning import PlDataModule
    except ImportError as e:  # the build image has neither lightning nor mfai
        raise SystemExit(
            f"bin/main.py needs the reference's `py4cast` package and `lightning` ({e}). "
            "Without them use py4cast_amd.trainer.Trainer (see README.md) or bench.py."
        )
    from py4cast_amd.lightning import AutoRegressiveLightning

    Py4castLightningCLI(AutoRegressiveLightning, PlDataModule)
