"""Import stand-in for OpenCV (see ../gym/__init__.py); nothing on the replay /
trainer path calls into it."""
INTER_AREA = 3
COLOR_RGB2GRAY = 7


def _absent(*a, **kw):
    raise RuntimeError("cv2 shim: not available")


resize = cvtColor = _absent


class ocl:
    @staticmethod
    def setUseOpenCL(flag):
        pass
