"""
Process-wide settings object. `from puzzlelib_amd.settings import Config` gives an object that is used exactly like
the reference's Config module (Config.py:9-56): Config.backend / Config.Backend.hip, Config.deviceIdx,
Config.allowMultiContext, Config.globalEvalMode, the check toggles, Config.getLogger(), Config.shouldInit().

This package serves one backend only — Backend.hip on an MI355X through libpuzzle_mi355.so. Selecting anything else
raises ConfigError when the device is first touched: there is no multi-backend dispatch and no CPU fallback.
"""
import sys, logging, multiprocessing
from enum import Enum


class ConfigError(Exception):
	pass


class Backend(Enum):
	cuda = 0
	hip = 1
	cpu = 2
	intel = 3


class Settings:
	Backend = Backend
	ConfigError = ConfigError

	def __init__(self):
		self.backend = Backend.hip
		self.deviceIdx = 0
		self.allowMultiContext = False

		self.globalEvalMode = False
		self.disableDtypeShapeChecks = False
		self.disableModuleCompatChecks = False
		self.verifyData = False
		self.showWarnings = True

		self.systemLog = False
		self.libname = "PuzzleLib"
		self.logger = None


	def shouldInit(self):
		return self.allowMultiContext or multiprocessing.current_process().name == "MainProcess"


	@staticmethod
	def isCPUBased(bnd):
		return bnd in (Backend.cpu, Backend.intel)


	def requireHip(self):
		if self.backend != Backend.hip:
			raise ConfigError("puzzlelib_amd implements Backend.hip (MI355X) only, got %s" % (self.backend, ))


	def getLogger(self):
		if self.logger is None:
			log = logging.getLogger(self.libname + ".mi355")
			log.setLevel(logging.DEBUG if self.systemLog else logging.INFO)
			log.propagate = False

			handler = logging.StreamHandler(stream=sys.stdout)
			handler.setFormatter(logging.Formatter("[" + self.libname + "] %(message)s"))
			log.addHandler(handler)

			self.logger = log

		return self.logger


Config = Settings()
