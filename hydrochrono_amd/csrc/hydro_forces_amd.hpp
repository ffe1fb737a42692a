// hydro_forces_amd.hpp -- umbrella over the C++ host-side mirror of the reference's plugin surface, which lives under
// include/hydroc_amd/ with the reference's own file layout (include/hydroc/*.h, src/hydro_types.h, src/hydro_yaml_parser.h,
// src/setup_hydro_from_yaml.h):
//   hydroc_amd/wave_types.h            WaveBase / NoWave / RegularWave / IrregularWaves(IrregularWaveParams)
//   hydroc_amd/hydro_forces.h          TestHydro, ComponentFunc, ForceFunc6d, HydroProfileStats (+ BodyView / MockBody for Chrono-free drivers)
//   hydroc_amd/chloadaddedmass.h       ChLoadAddedMass
//   hydroc_amd/hydro_types.h           HydroBody, WaveSettings, YAMLHydroData
//   hydroc_amd/hydro_yaml_parser.h     ReadHydroYAML
//   hydroc_amd/setup_hydro_from_yaml.h SetupHydroFromYAML
// Header-only; link with libhydrochrono_amd.so.  Kept under this name for programs written against rounds 1-3.
#pragma once

#include "../../include/hydroc_amd/hydro_forces.h"
#include "../../include/hydroc_amd/setup_hydro_from_yaml.h"
