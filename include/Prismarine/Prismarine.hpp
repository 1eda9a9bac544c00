#pragma once
// Prismarine/Prismarine.hpp -- umbrella include (reference Include/Prismarine/Prismarine.hpp:3-9)
#include "Utils.hpp"
#include "Structs.hpp"
#include "Radix.hpp"
#include "VertexInstance.hpp"
#include "TextureSet.hpp"
#include "MaterialSet.hpp"
#include "TriangleHierarchy.hpp"
#include "Pipeline.hpp"
#include "FrameBatch.hpp"   // addition: several frames in flight (psm_lanes_render)
