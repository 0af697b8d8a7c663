"""ilqr_amd: batched iLQR for AMD Instinct MI355X behind the malkstik/iLQR Python API."""
from .iLQRController import iLQR
from .batch import BatchSolver, BatchResult
from .models import Model, CarModel, Acrobot2DModel, QuadrotorEAModel, QuadrotorQuatModel

__all__ = ["iLQR", "BatchSolver", "BatchResult", "Model", "CarModel", "Acrobot2DModel", "QuadrotorEAModel",
           "QuadrotorQuatModel"]
__version__ = "0.1.0"
