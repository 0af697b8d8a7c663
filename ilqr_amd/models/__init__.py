from .model import Model
from .car import CarModel
from .acrobot import Acrobot2DModel
from .Quadrotor_EA import QuadrotorEAModel
from .Quadrotor_Quat import QuadrotorQuatModel
from .user import UserModel

__all__ = ["Model", "CarModel", "Acrobot2DModel", "QuadrotorEAModel", "QuadrotorQuatModel", "UserModel"]
