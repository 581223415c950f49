cmd = None
