"""bot7.models.abstract (models/abstract.lua:14-47): protocol only."""


class abstract(object):
    title = "bot7.models.abstract"

    def save(self):  # models/abstract.lua:20-21 (empty in the reference)
        pass

    def load(self):  # :23-24
        pass

    def update(self):  # :26-27
        pass

    def cache(self):  # :29-39
        return {"config": getattr(self, "config", None), "kernel": getattr(self, "kernel", None),
                "nzModel": getattr(self, "nzModel", None), "mean": getattr(self, "mean", None),
                "hyp": getattr(self, "hyp", None)}

    def class_(self):  # model:class() (:41-43); `class` is reserved in Python
        return self.title

    def __str__(self):
        return self.title


# config.sampler -> class (bots/bayesopt.lua:44).  The samplers are host code that stays in the reference's own Lua
# (samplers/slice.lua); whoever provides them registers them here (the test harness registers its stand-in).
sampler_registry = {}
