#pragma once
// Prismarine/Implementations.hpp -- inline implementations (reference Implementations.hpp:5-9)
#include "TriangleHierarchy.inl"
#include "Pipeline.inl"
