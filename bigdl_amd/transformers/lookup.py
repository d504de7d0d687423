def lookup_generate(self, *a, **k):  # replaced below in this round by the full implementation
    raise NotImplementedError
