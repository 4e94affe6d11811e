from .roboy_robot import RobotState, RoboyRobot
from .description import RobotDescription
from .msj_robot import MsjRobot, msj_platform_spec
from .upper_body_robot import UpperBodyRobot, upper_body_spec
