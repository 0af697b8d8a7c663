from ilqr_amd.models.acrobot import Acrobot2DModel  # noqa: F401
