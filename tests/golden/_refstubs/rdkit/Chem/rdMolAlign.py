CalcRMS = None
