run = None


def log(*a, **k):
    pass


def init(*a, **k):
    pass


def watch(*a, **k):
    pass
